// JSON input/output of the driver path:
//  * plonky3 proof JSON -> per-proof input vector.  Format = serde derive of the reference's
//    `Proof<Value<F>>` (/root/reference/src/p3/serde/proof.rs:16-19, 349-355); unknown keys such as
//    the artifact's "_marker" are ignored like serde does; order = `add_virtual_to` (proof.rs:357-373).
//  * flat plonky2 proof -> JSON in the shape of `serde_json::to_string(&ProofWithPublicInputs)`
//    (/root/reference/src/p3/mod.rs:261; field names per SURVEY.md App. A.10).
#include "json_io.h"
#include <ctype.h>
#include <stdexcept>
#include <string.h>

namespace p25 {
namespace {
struct JVal {
  enum Kind { NUL, NUM, STR, ARR, OBJ, BOOL } kind = NUL;
  u64 num = 0;
  std::vector<JVal> arr;
  std::vector<std::pair<std::string, JVal>> obj;
  const JVal& at(const char* key) const {
    for (auto& kv : obj)
      if (kv.first == key) return kv.second;
    throw std::invalid_argument(std::string("p3 proof JSON: missing key ") + key);
  }
};
struct Parser {
  const char* p;
  const char* e;
  int depth = 0;
  void ws() {
    while (p < e && isspace((unsigned char)*p)) p++;
  }
  [[noreturn]] void fail(const char* m) { throw std::invalid_argument(std::string("p3 proof JSON: ") + m); }
  JVal value() {
    if (++depth > 64) fail("nesting too deep");
    ws();
    if (p >= e) fail("unexpected end");
    JVal v;
    if (*p == '{') {
      v.kind = JVal::OBJ;
      p++;
      ws();
      if (p < e && *p == '}') {
        p++;
      } else {
        for (;;) {
          ws();
          JVal k = value();
          if (k.kind != JVal::STR) fail("object key must be a string");
          ws();
          if (p >= e || *p != ':') fail("expected ':'");
          p++;
          std::string key = k_str;
          JVal val = value();
          v.obj.emplace_back(std::move(key), std::move(val));
          ws();
          if (p < e && *p == ',') {
            p++;
            continue;
          }
          if (p < e && *p == '}') {
            p++;
            break;
          }
          fail("expected ',' or '}'");
        }
      }
    } else if (*p == '[') {
      v.kind = JVal::ARR;
      p++;
      ws();
      if (p < e && *p == ']') {
        p++;
      } else {
        for (;;) {
          v.arr.push_back(value());
          ws();
          if (p < e && *p == ',') {
            p++;
            continue;
          }
          if (p < e && *p == ']') {
            p++;
            break;
          }
          fail("expected ',' or ']'");
        }
      }
    } else if (*p == '"') {
      v.kind = JVal::STR;
      p++;
      k_str.clear();
      while (p < e && *p != '"') {
        if (*p == '\\') fail("escapes not supported in keys");
        k_str.push_back(*p++);
      }
      if (p >= e) fail("unterminated string");
      p++;
    } else if (isdigit((unsigned char)*p)) {
      v.kind = JVal::NUM;
      u64 x = 0;
      while (p < e && isdigit((unsigned char)*p)) {
        u64 d = (u64)(*p - '0');
        if (x > (~0ull - d) / 10) fail("number does not fit in u64");
        x = x * 10 + d;
        p++;
      }
      v.num = x;
    } else if (e - p >= 4 && !strncmp(p, "null", 4)) {
      p += 4;
    } else if (e - p >= 4 && !strncmp(p, "true", 4)) {
      v.kind = JVal::BOOL;
      v.num = 1;
      p += 4;
    } else if (e - p >= 5 && !strncmp(p, "false", 5)) {
      v.kind = JVal::BOOL;
      p += 5;
    } else {
      fail("unexpected character");
    }
    depth--;
    return v;
  }
  std::string k_str;  // last parsed string
};

u64 field(const JVal& v) {
  const JVal& x = v.at("value");
  if (x.kind != JVal::NUM) throw std::invalid_argument("p3 proof JSON: field element is not a number");
  if (x.num >= gl::P) throw std::invalid_argument("p3 proof JSON: non-canonical field element");
  return x.num;
}
void fields(const JVal& arr, std::vector<u64>& out) {
  for (auto& v : arr.arr) out.push_back(field(v));
}
void ext(const JVal& v, std::vector<u64>& out) { fields(v.at("value"), out); }
int log2_ceil(size_t n) {
  int r = 0;
  while (((size_t)1 << r) < n) r++;
  return r;
}
}  // namespace

void p3_proof_from_json(const char* json, size_t len, std::vector<u64>& inputs, P3Config& cfg) {
  Parser ps{json, json + len};
  JVal root = ps.value();
  ps.ws();
  if (ps.p != ps.e) throw std::invalid_argument("p3 proof JSON: trailing characters");
  inputs.clear();
  fields(root.at("commitments").at("trace").at("value"), inputs);
  fields(root.at("commitments").at("quotient_chunks").at("value"), inputs);
  const JVal& ov = root.at("opened_values");
  for (auto& e : ov.at("trace_local").arr) ext(e, inputs);
  for (auto& e : ov.at("trace_next").arr) ext(e, inputs);
  for (auto& chunk : ov.at("quotient_chunks").arr)
    for (auto& e : chunk.arr) ext(e, inputs);
  const JVal& fp = root.at("opening_proof").at("fri_proof");
  for (auto& c : fp.at("commit_phase_commits").arr) fields(c.at("value"), inputs);
  for (auto& qp : fp.at("query_proofs").arr)
    for (auto& step : qp.at("commit_phase_openings").arr) {
      ext(step.at("sibling_value"), inputs);
      for (auto& sib : step.at("opening_proof").arr) fields(sib, inputs);
    }
  ext(fp.at("final_poly"), inputs);
  inputs.push_back(field(fp.at("pow_witness")));
  const JVal& qo = root.at("opening_proof").at("query_openings");
  for (auto& q : qo.arr)
    for (auto& batch : q.arr) {
      for (auto& row : batch.at("opened_values").arr) fields(row, inputs);
      for (auto& sib : batch.at("opening_proof").arr) fields(sib, inputs);
    }
  // P3Config from the proof's shape (src/p3/mod.rs:74-87)
  if (qo.arr.empty() || qo.arr[0].arr.size() != 2) throw std::invalid_argument("p3 proof JSON: bad query_openings");
  cfg = P3Config();
  cfg.log_quotient_degree = log2_ceil(ov.at("quotient_chunks").arr.size());
  cfg.log_trace_height = (int)fp.at("commit_phase_commits").arr.size();
  cfg.trace_width = (int)ov.at("trace_local").arr.size();
  cfg.opening_matrix_log_max_height = (int)qo.arr[0].arr[0].at("opening_proof").arr.size();
  cfg.opening_proof_query_openings_opened_values_length = (int)qo.arr[0].arr[1].at("opened_values").arr.at(0).arr.size();
  cfg.degree_bits = (int)root.at("degree_bits").num;
  cfg.fri_config.num_queries = (int)fp.at("query_proofs").arr.size();
  // the FriConfig is not part of the proof (src/p3/mod.rs:242-246 passes it beside it); its log_blowup shows in the shape:
  // input Merkle paths are log_max_height = log_trace_height + log_blowup digests long (verifier.rs:264)
  cfg.fri_config.log_blowup = cfg.opening_matrix_log_max_height - cfg.log_trace_height;
  const size_t n_chunks = ov.at("quotient_chunks").arr.size();
  if (n_chunks != ((size_t)1 << cfg.log_quotient_degree) || n_chunks > 8)
    throw std::invalid_argument("p3 proof JSON: 1 quotient chunk (proof.rs:41-48), or 2 / 4 / 8 (AIRs of degree 3 / 4-5 / 6-9)");
  if (cfg.fri_config.log_blowup < 1 || cfg.fri_config.log_blowup > 4 || cfg.log_quotient_degree > cfg.fri_config.log_blowup)
    throw std::invalid_argument("p3 proof JSON: Merkle path lengths imply an unsupported log_blowup");
  if (inputs.size() != cfg.num_inputs()) throw std::invalid_argument("p3 proof JSON: shape is not rectangular");
}

// ---------------------------------------------------------------- proof -> JSON
namespace {
struct Out {
  std::string s;
  void num(u64 v) { s += std::to_string(v); }
  void hash(const u64* h) {
    s += "{\"elements\":[";
    for (int i = 0; i < 4; i++) {
      if (i) s += ',';
      num(h[i]);
    }
    s += "]}";
  }
  void hashes(const u64* h, size_t n) {
    s += '[';
    for (size_t i = 0; i < n; i++) {
      if (i) s += ',';
      hash(h + 4 * i);
    }
    s += ']';
  }
  void exts(const u64* e, size_t n) {
    s += '[';
    for (size_t i = 0; i < n; i++) {
      if (i) s += ',';
      s += '[';
      num(e[2 * i]);
      s += ',';
      num(e[2 * i + 1]);
      s += ']';
    }
    s += ']';
  }
  void nums(const u64* e, size_t n) {
    s += '[';
    for (size_t i = 0; i < n; i++) {
      if (i) s += ',';
      num(e[i]);
    }
    s += ']';
  }
};
}  // namespace

std::string proof_to_json(const Circuit& c, const ProofLayout& L, const u64* w) {
  Out o;
  const size_t capn = (size_t)1 << c.cfg.cap_height;
  const int NC = c.cfg.num_challenges;
  o.s += "{\"proof\":{\"wires_cap\":";
  o.hashes(w + L.wires_cap, capn);
  o.s += ",\"plonk_zs_partial_products_cap\":";
  o.hashes(w + L.zs_cap, capn);
  o.s += ",\"quotient_polys_cap\":";
  o.hashes(w + L.quotient_cap, capn);
  o.s += ",\"openings\":{\"constants\":";
  o.exts(w + L.constants, (L.sigmas - L.constants) / 2);
  o.s += ",\"plonk_sigmas\":";
  o.exts(w + L.sigmas, (L.wires - L.sigmas) / 2);
  o.s += ",\"wires\":";
  o.exts(w + L.wires, (L.zs - L.wires) / 2);
  o.s += ",\"plonk_zs\":";
  o.exts(w + L.zs, NC);
  o.s += ",\"plonk_zs_next\":";
  o.exts(w + L.zs_next, NC);
  o.s += ",\"partial_products\":";
  o.exts(w + L.pps, (L.quotient - L.pps) / 2);
  o.s += ",\"quotient_polys\":";
  o.exts(w + L.quotient, (L.fri_caps - L.quotient) / 2);
  o.s += ",\"lookup_zs\":[],\"lookup_zs_next\":[]},\"opening_proof\":{\"commit_phase_merkle_caps\":[";
  const size_t nl = c.fri_reduction_arity_bits.size();
  for (size_t l = 0; l < nl; l++) {
    if (l) o.s += ',';
    o.hashes(w + L.fri_caps + l * 4 * capn, capn);
  }
  o.s += "],\"query_round_proofs\":[";
  const int lde_bits = c.degree_bits + c.cfg.rate_bits;
  for (int q = 0; q < c.cfg.num_query_rounds; q++) {
    if (q) o.s += ',';
    const u64* p = w + L.queries + (size_t)q * L.query_stride;
    o.s += "{\"initial_trees_proof\":{\"evals_proofs\":[";
    for (int k = 0; k < 4; k++) {
      if (k) o.s += ',';
      o.s += '[';
      o.nums(p, L.oracle_width[k]);
      p += L.oracle_width[k];
      o.s += ",{\"siblings\":";
      o.hashes(p, lde_bits - c.cfg.cap_height);
      p += 4 * (size_t)(lde_bits - c.cfg.cap_height);
      o.s += "}]";
    }
    o.s += "]},\"steps\":[";
    int bits = lde_bits;
    for (size_t l = 0; l < nl; l++) {
      if (l) o.s += ',';
      const int ab = c.fri_reduction_arity_bits[l];
      bits -= ab;
      o.s += "{\"evals\":";
      o.exts(p, (size_t)1 << ab);
      p += 2 * ((size_t)1 << ab);
      o.s += ",\"merkle_proof\":{\"siblings\":";
      o.hashes(p, bits - c.cfg.cap_height);
      p += 4 * (size_t)(bits - c.cfg.cap_height);
      o.s += "}}";
    }
    o.s += "]}";
  }
  o.s += "],\"final_poly\":{\"coeffs\":";
  o.exts(w + L.final_poly, L.final_poly_len);
  o.s += "},\"pow_witness\":";
  o.num(w[L.pow_witness]);
  o.s += "}},\"public_inputs\":";
  o.nums(w + L.public_inputs, L.num_public_inputs);
  o.s += "}";
  return o.s;
}

// upstream util/serialization.rs `Buffer::write_proof_with_public_inputs` (what `ProofWithPublicInputs::to_bytes()`
// emits; absent crate, restated): every field element as 8 little-endian bytes in the order of the flat layout --
// caps, the opening set (constants, plonk_sigmas, wires, plonk_zs, plonk_zs_next, [lookup_zs, next_lookup_zs: empty],
// partial_products, quotient_polys), commit-phase caps, query rounds, final polynomial, PoW witness, public inputs
// -- with ONE difference: every Merkle proof is preceded by its sibling count as a u8 (write_merkle_proof).
namespace {
// calls f(is_path_len, value / count, n_words) over the proof in order
template <class F>
void walk_proof(const Circuit& c, const ProofLayout& L, F&& f) {
  const int lde_bits = c.degree_bits + c.cfg.rate_bits;
  f(false, L.queries);  // caps, openings, FRI caps: plain words
  for (int q = 0; q < c.cfg.num_query_rounds; q++) {
    for (int k = 0; k < 4; k++) {
      f(false, (size_t)L.oracle_width[k]);
      f(true, (size_t)(lde_bits - c.cfg.cap_height));
    }
    int bits = lde_bits;
    for (int ab : c.fri_reduction_arity_bits) {
      bits -= ab;
      f(false, 2 * ((size_t)1 << ab));
      f(true, (size_t)(bits - c.cfg.cap_height));
    }
  }
  f(false, L.total - L.final_poly);
}
}  // namespace
std::vector<uint8_t> proof_to_bytes(const Circuit& c, const ProofLayout& L, const u64* w) {
  std::vector<uint8_t> out;
  out.reserve(L.total * 8 + 256);
  size_t pos = 0;
  auto words = [&](size_t n) {
    for (size_t i = 0; i < n; i++) {
      u64 v = w[pos++];
      for (int b = 0; b < 8; b++) out.push_back((uint8_t)(v >> (8 * b)));
    }
  };
  walk_proof(c, L, [&](bool is_path, size_t n) {
    if (is_path) {
      if (n > 255) throw std::invalid_argument("Merkle proof length must fit in u8");
      out.push_back((uint8_t)n);
      words(4 * n);
    } else {
      words(n);
    }
  });
  if (pos != L.total) throw std::logic_error("proof_to_bytes: layout walk out of step");
  return out;
}
void proof_from_bytes(const Circuit& c, const ProofLayout& L, const uint8_t* data, size_t len, u64* w) {
  size_t pos = 0, off = 0;
  auto words = [&](size_t n) {
    if (off + 8 * n > len) throw std::invalid_argument("proof bytes truncated");
    for (size_t i = 0; i < n; i++) {
      u64 v = 0;
      for (int b = 0; b < 8; b++) v |= (u64)data[off + b] << (8 * b);
      if (v >= gl::P) throw std::invalid_argument("proof bytes: non-canonical field element");
      w[pos++] = v;
      off += 8;
    }
  };
  walk_proof(c, L, [&](bool is_path, size_t n) {
    if (is_path) {
      if (off >= len || data[off] != (uint8_t)n) throw std::invalid_argument("proof bytes: unexpected Merkle proof length");
      off++;
      words(4 * n);
    } else {
      words(n);
    }
  });
  if (off != len) throw std::invalid_argument("proof bytes: trailing data");
}

}  // namespace p25
