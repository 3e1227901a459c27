// Upstream's binary form of a built circuit: `CircuitData::to_bytes(&gate_serializer, &generator_serializer)` /
// `CircuitData::from_bytes` (plonky2 @ 3de92d9, util/serialization/mod.rs `write_circuit_data`; the crate is absent
// from /root/reference, so this is a restatement -- unpinned like the proof formats -- anchored on what the reference
// DOES hold: the `serialize` / `deserialize` bodies of its gates and generators, which fix the per-gate payloads and
// the signature generation of the API (`serialize(&self, dst: &mut Vec<u8>)`, no CommonCircuitData argument):
//   Poseidon2Gate        writes nothing                 /root/reference/src/common/poseidon2/poseidon2_gate.rs:399-405
//   Poseidon2Generator   writes row                     poseidon2_gate.rs:529-539
//   U32ArithmeticGate    writes num_ops                 src/common/u32/gates/arithmetic_u32.rs:287-300
//   U32ArithmeticGenerator  gate.serialize, row, i      arithmetic_u32.rs:445-464
//   U32InterleaveGate / Generator      num_ops / num_ops, row, i    interleave_u32.rs:237-247, 340-360
//   UninterleaveToU32Gate / Generator  num_ops / num_ops, row, i    uninterleave_to_u32.rs:272-283, 396-412
//
// Encoding (upstream `Write`): usize = u64 LE, u32 LE, bool = one byte, field element = canonical u64 LE, hash = 4
// field elements, Target = {true, row, column} | {false, index}, vectors = usize length + items except
// `write_field_vec` (no length).  A gate / generator is a u32 TAG -- its position in the serializer's type list --
// followed by its own `serialize` payload.  Tags: upstream's default lists (`default_gate_serializer`,
// `default_generator_serializer`, alphabetical) followed by the reference's types, i.e. the serializer a host declares as
//   impl_gate_serializer! { P25GateSerializer, <the 16 default gates>, Poseidon2Gate<F, D>, U32ArithmeticGate<F, D>,
//                           U32InterleaveGate, UninterleaveToU32Gate }
//   impl_generator_serializer! { P25GeneratorSerializer, <the 24 default generators>, Poseidon2Generator<F, D>,
//                                U32ArithmeticGenerator<F, D>, U32InterleaveGenerator, UninterleaveToU32Generator }
// (INTEGRATION.md section 5a).  Layout of the whole: CommonCircuitData | ProverOnlyCircuitData | VerifierOnlyCircuitData.
#include "circuit_bytes.h"
#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <string.h>
#include "circuit_io.h"

namespace p25 {
namespace {

// gate tags
enum : uint32_t {
  T_ARITHMETIC = 0, T_ARITH_EXT = 1, T_BASE_SUM = 2, T_CONSTANT = 3, T_COSET_INTERP = 4, T_EXPONENTIATION = 5, T_LOOKUP = 6,
  T_LOOKUP_TABLE = 7, T_MUL_EXT = 8, T_NOOP = 9, T_POSEIDON_MDS = 10, T_POSEIDON = 11, T_PUBLIC_INPUT = 12,
  T_RANDOM_ACCESS = 13, T_REDUCING_EXT = 14, T_REDUCING = 15, T_POSEIDON2 = 16, T_U32_ARITHMETIC = 17, T_U32_INTERLEAVE = 18,
  T_U32_UNINTERLEAVE = 19
};
// generator tags
enum : uint32_t {
  GT_ARITHMETIC_BASE = 0, GT_ARITH_EXT = 1, GT_BASE_SPLIT = 2, GT_BASE_SUM = 3, GT_CONSTANT = 4, GT_COPY = 5, GT_DUMMY_PROOF = 6,
  GT_EQUALITY = 7, GT_EXPONENTIATION = 8, GT_INTERPOLATION = 9, GT_LOOKUP = 10, GT_LOOKUP_TABLE = 11, GT_LOW_HIGH = 12,
  GT_MUL_EXT = 13, GT_NONZERO_TEST = 14, GT_POSEIDON = 15, GT_POSEIDON_MDS = 16, GT_QUOTIENT_EXT = 17, GT_RANDOM_ACCESS = 18,
  GT_RANDOM_VALUE = 19, GT_REDUCING = 20, GT_REDUCING_EXT = 21, GT_SPLIT = 22, GT_WIRE_SPLIT = 23, GT_POSEIDON2 = 24,
  GT_U32_ARITHMETIC = 25, GT_U32_INTERLEAVE = 26, GT_U32_UNINTERLEAVE = 27
};
struct GateTag {
  uint32_t tag;
  int has_param;   // 1: a single usize payload (num_ops / num_limbs / num_consts / num_power_bits / num_coeffs);
  u64 param;       // 3: RandomAccessGate's three (bits, num_copies, num_extra_constants)
};
GateTag gate_tag(GateKind k) {
  switch (k) {
    case G_NOOP: return {T_NOOP, 0, 0};
    case G_CONSTANT: return {T_CONSTANT, 1, 2};
    case G_PUBLIC_INPUT: return {T_PUBLIC_INPUT, 0, 0};
    case G_BASE_SUM: return {T_BASE_SUM, 1, (u64)BASE_SUM_LIMBS};
    case G_U32_INTERLEAVE: return {T_U32_INTERLEAVE, 1, (u64)gate_info(k).num_ops};
    case G_U32_UNINTERLEAVE: return {T_U32_UNINTERLEAVE, 1, (u64)gate_info(k).num_ops};
    case G_ARITHMETIC: return {T_ARITHMETIC, 1, (u64)gate_info(k).num_ops};
    case G_MUL_EXT: return {T_MUL_EXT, 1, (u64)gate_info(k).num_ops};
    case G_EXPONENTIATION: return {T_EXPONENTIATION, 1, (u64)EXP_POWER_BITS};
    case G_U32_ARITHMETIC: return {T_U32_ARITHMETIC, 1, (u64)gate_info(k).num_ops};
    case G_POSEIDON2: return {T_POSEIDON2, 0, 0};
    case G_ARITH_EXT: return {T_ARITH_EXT, 1, (u64)gate_info(k).num_ops};
    case G_POSEIDON: return {T_POSEIDON, 0, 0};
    case G_RANDOM_ACCESS: return {T_RANDOM_ACCESS, 3, (u64)RA_BITS};
    case G_REDUCING: return {T_REDUCING, 1, (u64)RED_COEFFS};
    case G_REDUCING_EXT: return {T_REDUCING_EXT, 1, (u64)REDX_COEFFS};
    case G_POSEIDON_MDS: return {T_POSEIDON_MDS, 0, 0};
    case G_COSET_INTERP: return {T_COSET_INTERP, 4, 4};   // subgroup_bits, degree, weights length, the 16 weights
    default: throw std::logic_error("gate kind without a serializer tag");
  }
}

struct W {
  std::vector<uint8_t> b;
  void raw(const void* p, size_t n) {
    size_t o = b.size();
    b.resize(o + n);
    if (n) memcpy(&b[o], p, n);
  }
  void u8(uint8_t v) { b.push_back(v); }
  void boolean(bool v) { u8(v ? 1 : 0); }
  void u32(uint32_t v) { raw(&v, 4); }
  void usize(u64 v) { raw(&v, 8); }
  void field(u64 v) { raw(&v, 8); }
  void fields(const u64* p, size_t n) { raw(p, n * 8); }
  void hash(const u64* h) { raw(h, 32); }
  void target(Target t) {
    if (t.is_virtual()) {
      boolean(false);
      usize((u64)t.col);
    } else {
      boolean(true);
      usize((u64)t.row);
      usize((u64)t.col);
    }
  }
  void target_vec(const std::vector<Target>& v) {
    usize(v.size());
    for (auto& t : v) target(t);
  }
};
struct R {
  const uint8_t* p;
  size_t len, off = 0;
  [[noreturn]] static void bad(const char* what) { throw std::invalid_argument(std::string("CircuitData bytes: ") + what); }
  void need(size_t n) {
    if (n > len - off) bad("truncated");
  }
  uint8_t u8() {
    need(1);
    return p[off++];
  }
  bool boolean() {
    uint8_t v = u8();
    if (v > 1) bad("bad bool");
    return v != 0;
  }
  uint32_t u32() {
    need(4);
    uint32_t v;
    memcpy(&v, p + off, 4);
    off += 4;
    return v;
  }
  u64 usize() {
    need(8);
    u64 v;
    memcpy(&v, p + off, 8);
    off += 8;
    return v;
  }
  u64 usize_max(u64 max, const char* what) {
    u64 v = usize();
    if (v > max) bad(what);
    return v;
  }
  u64 field() {
    u64 v = usize();
    if (v >= gl::P) bad("non-canonical field element");
    return v;
  }
  void fields(u64* out, size_t n) {
    need(n * 8);
    memcpy(out, p + off, n * 8);
    off += n * 8;
    for (size_t i = 0; i < n; i++)
      if (out[i] >= gl::P) bad("non-canonical field element");
  }
  void skip(size_t n) {
    need(n);
    off += n;
  }
  void hash(u64* h) { fields(h, 4); }
  Target target(u64 n_rows, u64 n_wires, u64 n_virtual) {
    if (boolean()) {
      u64 row = usize(), col = usize();
      if (row >= n_rows || col >= n_wires) bad("wire target out of range");
      return Target{(int32_t)row, (int32_t)col};
    }
    u64 idx = usize();
    if (idx >= n_virtual) bad("virtual target out of range");
    return Target{-1, (int32_t)idx};
  }
};

// fills upstream's MerkleTree::digests layout for the subtree under node (level, idx) of the level-ordered tree:
// left subtree's buffer | left child digest | right child digest | right subtree's buffer (a node is stored by its parent)
void fill_digests(const u64* tree, const std::vector<size_t>& level_off, int level, size_t idx, u64* buf) {
  if (level == 0) return;
  const size_t half = ((size_t)2 << (level - 1)) - 2;  // hashes in one child's buffer
  const u64* l = tree + level_off[level - 1] + 4 * (2 * idx);
  fill_digests(tree, level_off, level - 1, 2 * idx, buf);
  memcpy(buf + 4 * half, l, 32);
  memcpy(buf + 4 * (half + 1), l + 4, 32);
  fill_digests(tree, level_off, level - 1, 2 * idx + 1, buf + 4 * (half + 2));
}

void write_fri_config(W& w, const Circuit& c) {
  w.usize(c.cfg.rate_bits);
  w.usize(c.cfg.cap_height);
  w.usize(c.cfg.num_query_rounds);
  w.u32((uint32_t)c.cfg.proof_of_work_bits);
  w.u8(1);  // FriReductionStrategy::ConstantArityBits(arity_bits, final_poly_bits)
  w.usize(c.cfg.fri_arity_bits);
  w.usize(c.cfg.fri_final_poly_bits);
}
void read_fri_config(R& r, Circuit& c) {
  c.cfg.rate_bits = (int)r.usize_max(3, "rate_bits");
  c.cfg.cap_height = (int)r.usize_max(16, "cap_height");
  c.cfg.num_query_rounds = (int)r.usize_max(64, "num_query_rounds");
  c.cfg.proof_of_work_bits = (int)r.u32();
  if (c.cfg.proof_of_work_bits < 0 || c.cfg.proof_of_work_bits > 64) R::bad("proof_of_work_bits out of range");
  if (r.u8() != 1) R::bad("only FriReductionStrategy::ConstantArityBits is supported");
  c.cfg.fri_arity_bits = (int)r.usize_max(8, "arity bits");
  c.cfg.fri_final_poly_bits = (int)r.usize_max(32, "final poly bits");
}

void write_generator(W& w, const Circuit& c, const Generator& g) {
  auto row_of = [&](const Target& t) {
    if (t.is_virtual()) throw std::logic_error("gate generator on a virtual target");
    return (u64)t.row;
  };
  switch (g.kind) {
    case GEN_CONSTANT:
      w.u32(GT_CONSTANT);
      w.usize(row_of(g.outs[0]));
      w.usize((u64)g.outs[0].col);  // constant_index == wire_index in a ConstantGate
      w.usize((u64)g.outs[0].col);
      w.field(g.c0);
      break;
    case GEN_RANDOM:
      w.u32(GT_RANDOM_VALUE);
      w.target(g.outs[0]);
      break;
    case GEN_ARITHMETIC:
      w.u32(GT_ARITHMETIC_BASE);
      w.usize(row_of(g.outs[0]));
      w.field(g.c0);
      w.field(g.c1);
      w.usize((u64)g.outs[0].col / 4);
      break;
    case GEN_MUL_EXT:
      w.u32(GT_MUL_EXT);
      w.usize(row_of(g.outs[0]));
      w.field(g.c0);
      w.usize((u64)g.outs[0].col / 6);
      break;
    case GEN_ARITH_EXT:
      w.u32(GT_ARITH_EXT);
      w.usize(row_of(g.outs[0]));
      w.field(g.c0);
      w.field(g.c1);
      w.usize((u64)g.outs[0].col / 8);
      break;
    case GEN_QUOTIENT_EXT:
      w.u32(GT_QUOTIENT_EXT);
      for (int k = 0; k < 4; k++) w.target(g.deps[k]);  // numerator, denominator (ExtensionTarget = D targets)
      for (int k = 0; k < 2; k++) w.target(g.outs[k]);
      break;
    case GEN_BASE_SPLIT:
      w.u32(GT_BASE_SPLIT);
      w.usize(row_of(g.deps[0]));
      w.usize(g.outs.size());
      break;
    case GEN_WIRE_SPLIT:
      w.u32(GT_WIRE_SPLIT);
      w.target(g.deps[0]);
      w.usize(g.outs.size());
      for (auto& t : g.outs) w.usize(row_of(t));
      w.usize((u64)BASE_SUM_LIMBS);
      break;
    case GEN_BASE_SUM:
      w.u32(GT_BASE_SUM);
      w.usize(row_of(g.outs[0]));
      w.target_vec(g.deps);
      break;
    case GEN_LOW_HIGH:
      w.u32(GT_LOW_HIGH);
      w.target(g.deps[0]);
      w.usize((u64)g.aux);
      w.target(g.outs[0]);
      w.target(g.outs[1]);
      break;
    case GEN_EXPONENTIATION:
      w.u32(GT_EXPONENTIATION);
      w.usize(row_of(g.deps[0]));
      w.usize((u64)EXP_POWER_BITS);  // gate.serialize
      break;
    case GEN_POSEIDON:
    case GEN_POSEIDON2:
      w.u32(g.kind == GEN_POSEIDON ? GT_POSEIDON : GT_POSEIDON2);
      w.usize(row_of(g.deps[0]));
      break;
    case GEN_U32_ARITHMETIC:
      w.u32(GT_U32_ARITHMETIC);
      w.usize((u64)gate_info(G_U32_ARITHMETIC).num_ops);
      w.usize(row_of(g.deps[0]));
      w.usize((u64)g.deps[0].col / 6);
      break;
    case GEN_U32_INTERLEAVE:
      w.u32(GT_U32_INTERLEAVE);
      w.usize((u64)gate_info(G_U32_INTERLEAVE).num_ops);
      w.usize(row_of(g.deps[0]));
      w.usize((u64)g.deps[0].col / 2);
      break;
    case GEN_U32_UNINTERLEAVE:
      w.u32(GT_U32_UNINTERLEAVE);
      w.usize((u64)gate_info(G_U32_UNINTERLEAVE).num_ops);
      w.usize(row_of(g.deps[0]));
      w.usize((u64)g.deps[0].col / 3);
      break;
    case GEN_RANDOM_ACCESS:   // upstream RandomAccessGenerator { row, gate, copy }: row, copy, gate.serialize
      w.u32(GT_RANDOM_ACCESS);
      w.usize(row_of(g.deps[0]));
      w.usize((u64)g.deps[0].col / (2 + RA_VEC));
      w.usize(RA_BITS);
      w.usize(RA_COPIES);
      w.usize(RA_EXTRA_CONSTS);
      break;
    case GEN_POSEIDON_MDS:    // upstream PoseidonMdsGenerator { row }
      w.u32(GT_POSEIDON_MDS);
      w.usize(row_of(g.deps[0]));
      break;
    case GEN_COSET_INTERP:    // upstream InterpolationGenerator { row, gate }: row, gate.serialize
      w.u32(GT_INTERPOLATION);
      w.usize(row_of(g.deps[0]));
      w.usize(4);
      w.usize(CI_DEGREE);
      w.usize(CI_POINTS);
      {
        u64 x = 1;
        for (int i = 0; i < CI_POINTS; i++) {
          w.field(gl::mul(x, gl::inv(16)));
          x = gl::mul(x, gl::root_of_unity(4));
        }
      }
      break;
    case GEN_REDUCING:        // upstream ReducingGenerator { row, gate }: row, gate.serialize
    case GEN_REDUCING_EXT:
      w.u32(g.kind == GEN_REDUCING ? GT_REDUCING : GT_REDUCING_EXT);
      w.usize(row_of(g.deps[0]));
      w.usize(g.kind == GEN_REDUCING ? (u64)RED_COEFFS : (u64)REDX_COEFFS);
      break;
    default:
      throw std::logic_error("generator kind without a serializer tag");
  }
  (void)c;
}

}  // namespace

std::vector<uint8_t> circuit_data_to_bytes(const Circuit& c, const CircuitCommitment& cm) {
  W w;
  const size_t n = c.degree(), big = n << c.cfg.rate_bits;
  const size_t ncs = c.constants_sigmas.size(), RW = (size_t)c.cfg.num_routed_wires, Wn = (size_t)c.cfg.num_wires;
  // ---- CommonCircuitData
  w.usize(Wn);
  w.usize(RW);
  w.usize(c.cfg.num_constants);
  w.usize(100);  // security_bits
  w.usize(c.cfg.num_challenges);
  w.usize(c.cfg.max_quotient_degree_factor);
  w.boolean(true);   // use_base_arithmetic_gate
  w.boolean(false);  // zero_knowledge
  write_fri_config(w, c);
  // fri_params: config, reduction_arity_bits, degree_bits, hiding
  write_fri_config(w, c);
  w.usize(c.fri_reduction_arity_bits.size());
  for (int a : c.fri_reduction_arity_bits) w.usize((u64)a);
  w.usize(c.degree_bits);
  w.boolean(false);
  // selectors_info
  w.usize(c.selector_index.size());
  for (int s : c.selector_index) w.usize((u64)s);
  w.usize(c.groups.size());
  for (auto& gr : c.groups) {
    w.usize((u64)gr.first);
    w.usize((u64)gr.second);
  }
  w.usize(c.cfg.max_quotient_degree_factor);  // quotient_degree_factor
  w.usize(c.num_gate_constraints);
  w.usize(c.cfg.num_constants);
  w.usize(cm.public_inputs.size());
  w.usize(c.k_is.size());
  w.fields(c.k_is.data(), c.k_is.size());
  w.usize(c.num_partial_products);
  w.usize(0);  // num_lookup_polys
  w.usize(0);  // num_lookup_selectors
  w.usize(0);  // luts
  w.usize(c.gates.size());
  for (GateKind k : c.gates) {
    GateTag t = gate_tag(k);
    w.u32(t.tag);
    if (t.has_param) w.usize(t.param);
    if (t.has_param == 3) {
      w.usize(RA_COPIES);
      w.usize(RA_EXTRA_CONSTS);
    }
    if (t.has_param == 4) {  // CosetInterpolationGate::serialize: subgroup_bits, degree, barycentric_weights (len + values)
      w.usize(CI_DEGREE);
      w.usize(CI_POINTS);
      u64 x = 1;
      for (int i = 0; i < CI_POINTS; i++) {
        w.field(gl::mul(x, gl::inv(16)));
        x = gl::mul(x, gl::root_of_unity(4));
      }
    }
  }
  // ---- ProverOnlyCircuitData
  w.usize(c.generators.size());
  for (auto& g : c.generators) write_generator(w, c, g);
  {
    // generator_indices_by_watches: representative of every watched target -> the generators watching it (BTreeMap order)
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    for (size_t gi = 0; gi < c.generators.size(); gi++)
      for (auto& t : c.generators[gi].deps) pairs.push_back({c.rep[c.target_index(t)], (uint32_t)gi});
    std::sort(pairs.begin(), pairs.end());
    pairs.erase(std::unique(pairs.begin(), pairs.end()), pairs.end());
    size_t keys = 0;
    for (size_t i = 0; i < pairs.size(); i++) keys += i == 0 || pairs[i].first != pairs[i - 1].first;
    w.usize(keys);
    for (size_t i = 0; i < pairs.size();) {
      size_t j = i;
      while (j < pairs.size() && pairs[j].first == pairs[i].first) j++;
      w.usize(pairs[i].first);
      w.usize(j - i);
      for (size_t k = i; k < j; k++) w.usize(pairs[k].second);
      i = j;
    }
  }
  // constants_sigmas_commitment: PolynomialBatch { polynomials, merkle_tree, degree_log, rate_bits, blinding }
  w.usize(ncs);
  for (size_t p = 0; p < ncs; p++) {
    w.usize(n);
    w.fields(cm.coeffs + p * n, n);
  }
  w.usize(big);  // merkle_tree.leaves
  {
    std::vector<u64> leaf(ncs);
    for (size_t l = 0; l < big; l++) {
      for (size_t p = 0; p < ncs; p++) leaf[p] = cm.lde[p * big + l];
      w.usize(ncs);
      w.fields(leaf.data(), ncs);
    }
  }
  {
    const size_t cap_len = (size_t)1 << c.cfg.cap_height;
    if (cap_len > big) throw std::logic_error("cap larger than the tree");
    int levels = 0;
    while (((size_t)cap_len << levels) < big) levels++;
    std::vector<size_t> level_off(levels + 1);
    size_t o = 0, m = big;
    for (int l = 0; l <= levels; l++) {
      level_off[l] = o;
      o += 4 * m;
      m >>= 1;
    }
    const size_t per_cap = 2 * (big / cap_len) - 2;
    std::vector<u64> digests(4 * per_cap * cap_len);
    for (size_t ci = 0; ci < cap_len; ci++) fill_digests(cm.tree, level_off, levels, ci, digests.data() + 4 * per_cap * ci);
    w.usize(per_cap * cap_len);
    w.fields(digests.data(), digests.size());
    w.usize(c.cfg.cap_height);
    w.fields(cm.tree + level_off[levels], 4 * cap_len);
  }
  w.usize(c.degree_bits);
  w.usize(c.cfg.rate_bits);
  w.boolean(false);
  // sigmas (transposed: one vector of num_routed values per row), subgroup
  {
    const size_t s0 = ncs - RW;
    w.usize(n);
    std::vector<u64> rowv(RW);
    for (size_t r = 0; r < n; r++) {
      for (size_t j = 0; j < RW; j++) rowv[j] = c.constants_sigmas[s0 + j][r];
      w.usize(RW);
      w.fields(rowv.data(), RW);
    }
    w.usize(n);
    u64 g = gl::root_of_unity(c.degree_bits), x = 1;
    for (size_t i = 0; i < n; i++) {
      w.field(x);
      x = gl::mul(x, g);
    }
  }
  w.target_vec(cm.public_inputs);
  w.usize(c.rep.size());
  for (uint32_t r : c.rep) w.usize(r);
  {
    // fft_root_table(max_fft_points): row k = the first max(2^k, 2) powers of the primitive 2^(k+1)-th root
    const int lg = c.degree_bits + c.cfg.rate_bits;
    w.boolean(true);
    w.usize((u64)lg);
    for (int lg_m = 1; lg_m <= lg; lg_m++) {
      const size_t cnt = std::max<size_t>((size_t)1 << (lg_m - 1), 2);
      w.usize(cnt);
      u64 base = gl::root_of_unity(lg_m), x = 1;
      for (size_t i = 0; i < cnt; i++) {
        w.field(x);
        x = gl::mul(x, base);
      }
    }
  }
  w.hash(cm.digest);
  w.usize(0);  // lookup_rows
  w.usize(0);  // lut_to_lookups
  // ---- VerifierOnlyCircuitData
  w.usize(c.cfg.cap_height);
  {
    const size_t cap_len = (size_t)1 << c.cfg.cap_height;
    size_t o = 0;
    for (size_t m = big; m > cap_len; m >>= 1) o += 4 * m;
    w.fields(cm.tree + o, 4 * cap_len);
  }
  w.hash(cm.digest);
  return std::move(w.b);
}

namespace {
// coefficients -> values on the subgroup (host, once per import): iterative radix-2 DIT
void ntt_forward_host(std::vector<u64>& a, int log_n) {
  const size_t n = a.size();
  for (size_t i = 0; i < n; i++) {
    size_t j = gl::bitrev((u32)i, (unsigned)log_n);
    if (i < j) std::swap(a[i], a[j]);
  }
  for (int s = 1; s <= log_n; s++) {
    const size_t m = (size_t)1 << s, h = m >> 1;
    const u64 wm = gl::root_of_unity((unsigned)s);
    std::vector<u64> tw(h);
    u64 x = 1;
    for (size_t j = 0; j < h; j++) {
      tw[j] = x;
      x = gl::mul(x, wm);
    }
    for (size_t k = 0; k < n; k += m)
      for (size_t j = 0; j < h; j++) {
        u64 t = gl::mul(tw[j], a[k + j + h]), u = a[k + j];
        a[k + j] = gl::add(u, t);
        a[k + j + h] = gl::sub(u, t);
      }
  }
}
}  // namespace

Circuit circuit_data_from_bytes(const uint8_t* data, size_t len, const uint32_t* input_target_indices, size_t n_inputs,
                                u64 digest_out[4]) {
  R r{data, len};
  Circuit c;
  // ---- CommonCircuitData
  c.cfg.num_wires = (int)r.usize_max(1024, "num_wires");
  c.cfg.num_routed_wires = (int)r.usize_max(1024, "num_routed_wires");
  c.cfg.num_constants = (int)r.usize_max(64, "num_constants");
  r.usize();  // security_bits
  c.cfg.num_challenges = (int)r.usize_max(8, "num_challenges");
  c.cfg.max_quotient_degree_factor = (int)r.usize_max(64, "max_quotient_degree_factor");
  r.boolean();  // use_base_arithmetic_gate
  if (r.boolean()) R::bad("zero_knowledge circuits are not supported");
  read_fri_config(r, c);
  {
    Circuit tmp;
    read_fri_config(r, tmp);  // fri_params.config (the same values again)
    if (tmp.cfg.rate_bits != c.cfg.rate_bits || tmp.cfg.cap_height != c.cfg.cap_height) R::bad("fri_params disagree with the config");
  }
  {
    u64 nl = r.usize_max(8, "too many FRI layers");
    for (u64 i = 0; i < nl; i++) c.fri_reduction_arity_bits.push_back((int)r.usize_max(8, "FRI arity bits"));
  }
  c.degree_bits = (int)r.usize_max(22, "degree_bits");
  if (c.degree_bits < 1) R::bad("degree_bits");
  if (r.boolean()) R::bad("hiding FRI is not supported");
  const size_t n = c.degree(), big = n << c.cfg.rate_bits;
  const size_t Wn = (size_t)c.cfg.num_wires, RW = (size_t)c.cfg.num_routed_wires;
  if (Wn < 1 || RW < 1 || RW > Wn) R::bad("wire counts");
  {
    u64 ng = r.usize_max(16, "gate count");
    for (u64 i = 0; i < ng; i++) c.selector_index.push_back((int)r.usize_max(15, "selector index"));
    u64 ngr = r.usize_max(16, "group count");
    for (u64 i = 0; i < ngr; i++) {
      u64 a = r.usize_max(16, "group"), b = r.usize_max(16, "group");
      c.groups.push_back({(int)a, (int)b});
    }
    c.num_selectors = (int)ngr;
  }
  if (r.usize() != (u64)c.cfg.max_quotient_degree_factor) R::bad("quotient_degree_factor != max_quotient_degree_factor");
  c.num_gate_constraints = (int)r.usize_max(1 << 20, "num_gate_constraints");
  if (r.usize() != (u64)c.cfg.num_constants) R::bad("num_constants disagree");
  const u64 n_pi = r.usize_max(MAX_PUBLIC_INPUTS, "num_public_inputs");
  if (r.usize() != RW) R::bad("k_is length");
  c.k_is.resize(RW);
  r.fields(c.k_is.data(), RW);
  c.num_partial_products = (int)r.usize_max(1 << 20, "num_partial_products");
  if (r.usize() != 0 || r.usize() != 0 || r.usize() != 0) R::bad("lookup tables are not supported");
  {
    u64 ng = r.usize();
    if (ng != c.selector_index.size()) R::bad("gate list length != selector_indices length");
    for (u64 i = 0; i < ng; i++) {
      const uint32_t tag = r.u32();
      GateKind kind = G_NUM_KINDS;
      for (int k = 0; k < G_NUM_KINDS; k++)
        if (gate_tag((GateKind)k).tag == tag) kind = (GateKind)k;
      if (kind == G_NUM_KINDS) R::bad("a gate type this library has no evaluator for");
      GateTag t = gate_tag(kind);
      if (t.has_param && r.usize() != t.param) R::bad("a gate with parameters this library does not support");
      if (t.has_param == 3 && (r.usize() != (u64)RA_COPIES || r.usize() != (u64)RA_EXTRA_CONSTS))
        R::bad("a RandomAccessGate other than new_from_config(standard_recursion_config, 4)");
      if (t.has_param == 4) {
        if (r.usize() != (u64)CI_DEGREE || r.usize() != (u64)CI_POINTS) R::bad("a CosetInterpolationGate other than with_max_degree(4, 8)");
        u64 x = 1;
        for (int k = 0; k < CI_POINTS; k++) {
          if (r.field() != gl::mul(x, gl::inv(16))) R::bad("CosetInterpolationGate barycentric weights");
          x = gl::mul(x, gl::root_of_unity(4));
        }
      }
      c.gates.push_back(kind);
    }
  }
  // ---- ProverOnlyCircuitData: generators are expanded once the representative map (hence the number of virtual
  // targets) is known; remember where they start
  const size_t gens_at = r.off;
  const u64 n_gens = r.usize_max((u64)1 << 28, "generator count");
  // first pass over the generators: skip (the payload sizes are fixed per tag except for the vectors)
  auto skip_target = [&]() {
    if (r.boolean()) r.skip(16); else r.skip(8);
  };
  for (u64 i = 0; i < n_gens; i++) {
    switch (r.u32()) {
      case GT_CONSTANT: r.skip(32); break;
      case GT_RANDOM_VALUE: skip_target(); break;
      case GT_ARITHMETIC_BASE: case GT_ARITH_EXT: r.skip(32); break;
      case GT_MUL_EXT: r.skip(24); break;
      case GT_QUOTIENT_EXT: for (int k = 0; k < 6; k++) skip_target(); break;
      case GT_BASE_SPLIT: r.skip(16); break;
      case GT_WIRE_SPLIT: { skip_target(); u64 k = r.usize_max(64, "WireSplitGenerator gates"); r.skip(8 * k + 8); break; }
      case GT_BASE_SUM: { r.skip(8); u64 k = r.usize_max(64, "BaseSumGenerator limbs"); for (u64 j = 0; j < k; j++) skip_target(); break; }
      case GT_LOW_HIGH: skip_target(); r.skip(8); skip_target(); skip_target(); break;
      case GT_EXPONENTIATION: r.skip(16); break;
      case GT_POSEIDON: case GT_POSEIDON2: case GT_POSEIDON_MDS: r.skip(8); break;
      case GT_U32_ARITHMETIC: case GT_U32_INTERLEAVE: case GT_U32_UNINTERLEAVE: r.skip(24); break;
      case GT_RANDOM_ACCESS: r.skip(40); break;
      case GT_REDUCING: case GT_REDUCING_EXT: r.skip(16); break;
      case GT_INTERPOLATION: r.skip(8 + 24 + 8 * CI_POINTS); break;
      default: R::bad("a generator type this library has no body for");
    }
  }
  {
    u64 keys = r.usize_max((u64)1 << 31, "watch map");
    for (u64 i = 0; i < keys; i++) {
      r.usize();
      u64 k = r.usize_max((u64)1 << 31, "watch list");
      r.skip(8 * k);
    }
  }
  // constants_sigmas_commitment
  const u64 ncs = r.usize_max(4096, "polynomial count");
  if (ncs != (u64)c.num_selectors + c.cfg.num_constants + RW) R::bad("constants_sigmas count != selectors + constants + routed wires");
  // sizes come from untrusted header fields: the input must hold that much data BEFORE anything is allocated for it
  if ((len - r.off) / (8 + 8 * (u64)n) < ncs) R::bad("truncated (constants/sigmas polynomials)");
  c.constants_sigmas.assign(ncs, std::vector<u64>(n));
  for (u64 p = 0; p < ncs; p++) {
    if (r.usize() != n) R::bad("polynomial length != 2^degree_bits");
    r.fields(c.constants_sigmas[p].data(), n);
    ntt_forward_host(c.constants_sigmas[p], c.degree_bits);  // the batch stores coefficients; the prover's tables are values
  }
  if (r.usize() != big) R::bad("leaf count != 2^(degree_bits + rate_bits)");
  for (size_t l = 0; l < big; l++) {
    if (r.usize() != ncs) R::bad("leaf width");
    r.skip(8 * ncs);  // recomputed on the device (and checked through the digest)
  }
  {
    u64 nd = r.usize_max((u64)1 << 32, "digest count");
    r.skip(32 * nd);
    if (r.usize() != (u64)c.cfg.cap_height) R::bad("cap height");
    r.skip(32 * ((size_t)1 << c.cfg.cap_height));
  }
  if (r.usize() != (u64)c.degree_bits || r.usize() != (u64)c.cfg.rate_bits || r.boolean()) R::bad("polynomial batch shape");
  {
    if (r.usize() != n) R::bad("sigmas length");
    const size_t s0 = ncs - RW;
    std::vector<u64> rowv(RW);
    for (size_t row = 0; row < n; row++) {
      if (r.usize() != RW) R::bad("sigma row length");
      r.fields(rowv.data(), RW);
      for (size_t j = 0; j < RW; j++)
        if (rowv[j] != c.constants_sigmas[s0 + j][row]) R::bad("sigmas disagree with the committed sigma polynomials");
    }
    if (r.usize() != n) R::bad("subgroup length");
    r.skip(8 * n);
  }
  const size_t pis_at = r.off;   // the targets are range-checked once the number of virtual targets is known
  if (r.usize() != n_pi) R::bad("public_inputs length != num_public_inputs");
  for (u64 i = 0; i < n_pi; i++) {
    if (r.boolean()) r.skip(16); else r.skip(8);
  }
  {
    u64 nt = r.usize_max((u64)1 << 31, "representative map");
    if (nt < n * Wn) R::bad("representative map shorter than the wire grid");
    c.num_virtual_targets = nt - n * Wn;
    if ((len - r.off) / 8 < nt) R::bad("truncated (representative map)");
    c.rep.resize(nt);
    for (u64 i = 0; i < nt; i++) {
      u64 v = r.usize();
      if (v >= nt) R::bad("representative out of range");
      c.rep[i] = (uint32_t)v;
    }
  }
  if (r.boolean()) {
    u64 rows = r.usize_max(64, "fft root table");
    for (u64 i = 0; i < rows; i++) {
      u64 k = r.usize_max((u64)1 << 32, "fft root table row");
      r.skip(8 * k);
    }
  }
  r.hash(digest_out);
  if (r.usize() != 0 || r.usize() != 0) R::bad("lookup tables are not supported");
  // ---- VerifierOnlyCircuitData
  if (r.usize() != (u64)c.cfg.cap_height) R::bad("verifier cap height");
  r.skip(32 * ((size_t)1 << c.cfg.cap_height));
  {
    u64 d2[4];
    r.hash(d2);
    if (memcmp(d2, digest_out, 32)) R::bad("prover and verifier circuit digests differ");
  }
  if (r.off != len) R::bad("trailing bytes");

  // rows: the gate of a row is what its selector polynomial says; its constants are the constant polynomials
  c.rows.resize(n);
  for (size_t row = 0; row < n; row++) {
    int found = -1;
    for (int s = 0; s < c.num_selectors; s++) {
      const u64 v = c.constants_sigmas[s][row];
      if (v == 0xFFFFFFFFull) continue;  // UNUSED_SELECTOR
      if (v >= c.gates.size() || c.selector_index[v] != s || found >= 0) R::bad("selector polynomials do not single out one gate per row");
      found = (int)v;
    }
    if (found < 0) R::bad("row without a gate");
    c.rows[row].kind = c.gates[found];
    c.rows[row].constants[0] = c.constants_sigmas[c.num_selectors][row];
    c.rows[row].constants[1] = c.cfg.num_constants > 1 ? c.constants_sigmas[c.num_selectors + 1][row] : 0;
    if (c.rows[row].kind == G_PUBLIC_INPUT) c.pi_row = (int)row;
  }
  // generators, second pass
  {
    R g{data, len};
    g.off = gens_at;
    g.usize();
    const u64 nv = c.num_virtual_targets;
    auto tgt = [&]() { return g.target(n, Wn, nv); };
    auto row_kind = [&](u64 row, GateKind k, const char* what) {
      if (row >= n || c.rows[row].kind != k) R::bad(what);
      return (int)row;
    };
    c.generators.reserve(n_gens);
    for (u64 i = 0; i < n_gens; i++) {
      Generator gen;
      switch (g.u32()) {
        case GT_CONSTANT: {
          int row = row_kind(g.usize(), G_CONSTANT, "ConstantGenerator outside a ConstantGate");
          u64 ci = g.usize(), wi = g.usize();
          if (ci != wi || wi >= 2) R::bad("ConstantGenerator indices");
          gen.kind = GEN_CONSTANT;
          gen.c0 = g.field();
          gen.outs = {wire(row, (int)wi)};
          break;
        }
        case GT_RANDOM_VALUE: {
          gen.kind = GEN_RANDOM;
          Target t = tgt();
          gen.aux = t.col;
          gen.outs = {t};
          break;
        }
        case GT_ARITHMETIC_BASE: {
          int row = row_kind(g.usize(), G_ARITHMETIC, "ArithmeticBaseGenerator outside an ArithmeticGate");
          u64 k[2] = {g.field(), g.field()};
          u64 op = g.usize_max((u64)gate_info(G_ARITHMETIC).num_ops - 1, "op index");
          gen = gate_op_generator(G_ARITHMETIC, k, row, (int)op);
          break;
        }
        case GT_ARITH_EXT: {
          int row = row_kind(g.usize(), G_ARITH_EXT, "ArithmeticExtensionGenerator outside its gate");
          u64 k[2] = {g.field(), g.field()};
          u64 op = g.usize_max((u64)gate_info(G_ARITH_EXT).num_ops - 1, "op index");
          gen = gate_op_generator(G_ARITH_EXT, k, row, (int)op);
          break;
        }
        case GT_MUL_EXT: {
          int row = row_kind(g.usize(), G_MUL_EXT, "MulExtensionGenerator outside its gate");
          u64 k[2] = {g.field(), 0};
          u64 op = g.usize_max((u64)gate_info(G_MUL_EXT).num_ops - 1, "op index");
          gen = gate_op_generator(G_MUL_EXT, k, row, (int)op);
          break;
        }
        case GT_QUOTIENT_EXT:
          gen.kind = GEN_QUOTIENT_EXT;
          for (int k = 0; k < 4; k++) gen.deps.push_back(tgt());
          for (int k = 0; k < 2; k++) gen.outs.push_back(tgt());
          break;
        case GT_BASE_SPLIT: {
          int row = row_kind(g.usize(), G_BASE_SUM, "BaseSplitGenerator outside a BaseSumGate");
          if (g.usize() != (u64)BASE_SUM_LIMBS) R::bad("BaseSplitGenerator limbs");
          u64 k[2] = {0, 0};
          gen = gate_op_generator(G_BASE_SUM, k, row, 0);
          break;
        }
        case GT_WIRE_SPLIT: {
          gen.kind = GEN_WIRE_SPLIT;
          gen.deps = {tgt()};
          u64 k = g.usize_max(2, "WireSplitGenerator gates");
          for (u64 j = 0; j < k; j++) gen.outs.push_back(wire(row_kind(g.usize(), G_BASE_SUM, "WireSplitGenerator gate"), 0));
          if (g.usize() != (u64)BASE_SUM_LIMBS) R::bad("WireSplitGenerator limbs");
          break;
        }
        case GT_BASE_SUM: {
          int row = row_kind(g.usize(), G_BASE_SUM, "BaseSumGenerator outside a BaseSumGate");
          gen.kind = GEN_BASE_SUM;
          u64 k = g.usize_max(64, "BaseSumGenerator limbs");
          for (u64 j = 0; j < k; j++) gen.deps.push_back(tgt());
          gen.outs = {wire(row, 0)};
          break;
        }
        case GT_LOW_HIGH:
          gen.kind = GEN_LOW_HIGH;
          gen.deps = {tgt()};
          gen.aux = (int)g.usize_max(63, "n_log");
          gen.outs = {tgt(), tgt()};
          break;
        case GT_EXPONENTIATION: {
          int row = row_kind(g.usize(), G_EXPONENTIATION, "ExponentiationGenerator outside its gate");
          if (g.usize() != (u64)EXP_POWER_BITS) R::bad("ExponentiationGate power bits");
          u64 k[2] = {0, 0};
          gen = gate_op_generator(G_EXPONENTIATION, k, row, 0);
          break;
        }
        case GT_POSEIDON: {
          u64 k[2] = {0, 0};
          gen = gate_op_generator(G_POSEIDON, k, row_kind(g.usize(), G_POSEIDON, "PoseidonGenerator outside its gate"), 0);
          break;
        }
        case GT_POSEIDON2: {
          u64 k[2] = {0, 0};
          gen = gate_op_generator(G_POSEIDON2, k, row_kind(g.usize(), G_POSEIDON2, "Poseidon2Generator outside its gate"), 0);
          break;
        }
        case GT_U32_ARITHMETIC:
        case GT_U32_INTERLEAVE:
        case GT_U32_UNINTERLEAVE: {
          g.off -= 4;
          const uint32_t tag = g.u32();
          const GateKind gk = tag == GT_U32_ARITHMETIC ? G_U32_ARITHMETIC : (tag == GT_U32_INTERLEAVE ? G_U32_INTERLEAVE : G_U32_UNINTERLEAVE);
          if (g.usize() != (u64)gate_info(gk).num_ops) R::bad("u32 gate num_ops");
          int row = row_kind(g.usize(), gk, "u32 generator outside its gate");
          u64 op = g.usize_max((u64)gate_info(gk).num_ops - 1, "op index");
          u64 k[2] = {0, 0};
          gen = gate_op_generator(gk, k, row, (int)op);
          break;
        }
        case GT_RANDOM_ACCESS: {
          int row = row_kind(g.usize(), G_RANDOM_ACCESS, "RandomAccessGenerator outside its gate");
          u64 copy = g.usize_max(RA_COPIES - 1, "copy index");
          if (g.usize() != (u64)RA_BITS || g.usize() != (u64)RA_COPIES || g.usize() != (u64)RA_EXTRA_CONSTS) R::bad("RandomAccessGate shape");
          u64 k[2] = {0, 0};
          gen = gate_op_generator(G_RANDOM_ACCESS, k, row, (int)copy);
          break;
        }
        case GT_POSEIDON_MDS: {
          u64 k[2] = {0, 0};
          gen = gate_op_generator(G_POSEIDON_MDS, k, row_kind(g.usize(), G_POSEIDON_MDS, "PoseidonMdsGenerator outside its gate"), 0);
          break;
        }
        case GT_INTERPOLATION: {
          int row = row_kind(g.usize(), G_COSET_INTERP, "InterpolationGenerator outside its gate");
          if (g.usize() != 4 || g.usize() != (u64)CI_DEGREE || g.usize() != (u64)CI_POINTS) R::bad("CosetInterpolationGate shape");
          g.skip(8 * CI_POINTS);   // the weights were checked with the gate list
          u64 k[2] = {0, 0};
          gen = gate_op_generator(G_COSET_INTERP, k, row, 0);
          break;
        }
        case GT_REDUCING:
        case GT_REDUCING_EXT: {
          g.off -= 4;
          const bool ext = g.u32() == GT_REDUCING_EXT;
          const GateKind gk = ext ? G_REDUCING_EXT : G_REDUCING;
          int row = row_kind(g.usize(), gk, "ReducingGenerator outside its gate");
          if (g.usize() != (u64)(ext ? REDX_COEFFS : RED_COEFFS)) R::bad("reducing gate num_coeffs");
          u64 k[2] = {0, 0};
          gen = gate_op_generator(gk, k, row, 0);
          break;
        }
        default:
          R::bad("a generator type this library has no body for");
      }
      c.generators.push_back(std::move(gen));
    }
  }
  {
    R g{data, len};
    g.off = pis_at;
    g.usize();
    for (u64 i = 0; i < n_pi; i++) c.public_inputs.push_back(g.target(n, Wn, c.num_virtual_targets));
  }
  // the per-proof inputs are not part of CircuitData (upstream hands a PartialWitness to prove): the caller names them
  const size_t NT = c.num_targets();
  for (size_t i = 0; i < n_inputs; i++) {
    const uint32_t idx = input_target_indices[i];
    if (idx >= NT) R::bad("input target out of range");
    c.input_targets.push_back(idx >= n * Wn ? Target{-1, (int32_t)(idx - n * Wn)} : Target{(int32_t)(idx / Wn), (int32_t)(idx % Wn)});
  }
  // everything else (kernel capacities, group / selector consistency, generator shapes) is checked where every
  // circuit enters the library
  std::vector<uint8_t> blob = circuit_to_blob(c);
  return circuit_from_blob(blob.data(), blob.size());
}

}  // namespace p25
