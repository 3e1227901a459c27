#include "circuit_io.h"
#include <stdexcept>
#include <string.h>
namespace p25 {
namespace {
struct Writer {
  std::vector<uint8_t> b;
  void u64w(u64 v) { size_t o = b.size(); b.resize(o + 8); memcpy(&b[o], &v, 8); }
  void u32arr(const uint32_t* p, size_t n) {
    size_t o = b.size(), bytes = (n * 4 + 7) & ~(size_t)7;
    b.resize(o + bytes, 0);
    if (n) memcpy(&b[o], p, n * 4);
  }
  void u64arr(const u64* p, size_t n) { size_t o = b.size(); b.resize(o + n * 8); if (n) memcpy(&b[o], p, n * 8); }
};
struct Reader {
  const uint8_t* p; size_t len, off = 0;
  u64 u64r() { if (off + 8 > len) throw std::invalid_argument("circuit blob truncated"); u64 v; memcpy(&v, p + off, 8); off += 8; return v; }
  void u32arr(uint32_t* out, size_t n) {
    size_t bytes = (n * 4 + 7) & ~(size_t)7;
    if (off + bytes > len) throw std::invalid_argument("circuit blob truncated");
    if (n) memcpy(out, p + off, n * 4);
    off += bytes;
  }
  void u64arr(u64* out, size_t n) { if (off + n * 8 > len) throw std::invalid_argument("circuit blob truncated"); if (n) memcpy(out, p + off, n * 8); off += n * 8; }
};
const u64 MAGIC = 0x3143524943353250ull;  // "P25CIRC1"
}  // namespace

std::vector<uint8_t> circuit_to_blob(const Circuit& c) {
  Writer w;
  const size_t n = c.degree();
  w.u64w(MAGIC);
  u64 h[32] = {0};
  h[0] = c.degree_bits; h[1] = c.cfg.num_wires; h[2] = c.cfg.num_routed_wires; h[3] = c.cfg.num_constants;
  h[4] = c.cfg.num_challenges; h[5] = c.cfg.max_quotient_degree_factor; h[6] = c.cfg.rate_bits;
  h[7] = c.cfg.cap_height; h[8] = c.cfg.proof_of_work_bits; h[9] = c.cfg.num_query_rounds;
  h[10] = c.fri_reduction_arity_bits.size(); h[11] = c.num_selectors; h[12] = c.num_gate_constraints;
  h[13] = c.num_partial_products; h[14] = c.gates.size(); h[15] = (u64)c.pi_row; h[16] = c.num_virtual_targets;
  h[17] = c.input_targets.size(); h[18] = c.generators.size(); h[19] = c.constants_sigmas.size();
  h[20] = c.cfg.fri_arity_bits; h[21] = c.cfg.fri_final_poly_bits;
  w.u64arr(h, 32);
  for (size_t i = 0; i < c.gates.size(); i++) {
    int s = c.selector_index[i];
    w.u64w(c.gates[i]); w.u64w(s); w.u64w(c.groups[s].first); w.u64w(c.groups[s].second);
  }
  for (int a : c.fri_reduction_arity_bits) w.u64w(a);
  std::vector<uint32_t> kinds(n);
  for (size_t i = 0; i < n; i++) kinds[i] = c.rows[i].kind;
  w.u32arr(kinds.data(), n);
  for (auto& p : c.constants_sigmas) w.u64arr(p.data(), n);
  w.u64arr(c.k_is.data(), c.k_is.size());
  std::vector<uint32_t> in(c.input_targets.size());
  for (size_t i = 0; i < in.size(); i++) in[i] = (uint32_t)c.target_index(c.input_targets[i]);
  w.u32arr(in.data(), in.size());
  w.u32arr(c.rep.data(), c.rep.size());
  std::vector<uint32_t> args;
  for (auto& g : c.generators) {
    w.u64w(g.kind); w.u64w(g.c0); w.u64w(g.c1); w.u64w((u64)g.aux); w.u64w(g.deps.size()); w.u64w(g.outs.size());
    args.clear();
    for (auto& t : g.deps) args.push_back((uint32_t)c.target_index(t));
    for (auto& t : g.outs) args.push_back((uint32_t)c.target_index(t));
    w.u32arr(args.data(), args.size());
  }
  return std::move(w.b);
}

Circuit circuit_from_blob(const uint8_t* data, size_t len) {
  Reader r{data, len};
  if (r.u64r() != MAGIC) throw std::invalid_argument("not a circuit blob");
  u64 h[32];
  r.u64arr(h, 32);
  Circuit c;
  c.degree_bits = (int)h[0]; c.cfg.num_wires = (int)h[1]; c.cfg.num_routed_wires = (int)h[2]; c.cfg.num_constants = (int)h[3];
  c.cfg.num_challenges = (int)h[4]; c.cfg.max_quotient_degree_factor = (int)h[5]; c.cfg.rate_bits = (int)h[6];
  c.cfg.cap_height = (int)h[7]; c.cfg.proof_of_work_bits = (int)h[8]; c.cfg.num_query_rounds = (int)h[9];
  size_t n_arity = h[10]; c.num_selectors = (int)h[11]; c.num_gate_constraints = (int)h[12];
  c.num_partial_products = (int)h[13]; size_t ng = h[14]; c.pi_row = (int)h[15]; c.num_virtual_targets = h[16];
  size_t n_in = h[17], n_gen = h[18], n_cs = h[19];
  c.cfg.fri_arity_bits = (int)h[20]; c.cfg.fri_final_poly_bits = (int)h[21];
  if (c.degree_bits > 24 || ng > G_NUM_KINDS || n_cs > 4096) throw std::invalid_argument("circuit blob: bad header");
  const size_t n = c.degree();
  const int W = c.cfg.num_wires;
  c.groups.assign(c.num_selectors, {0, 0});
  for (size_t i = 0; i < ng; i++) {
    u64 k = r.u64r(), s = r.u64r(), gs = r.u64r(), ge = r.u64r();
    if (k >= G_NUM_KINDS || s >= (u64)c.num_selectors) throw std::invalid_argument("circuit blob: bad gate");
    c.gates.push_back((GateKind)k); c.selector_index.push_back((int)s); c.groups[s] = {(int)gs, (int)ge};
  }
  for (size_t i = 0; i < n_arity; i++) c.fri_reduction_arity_bits.push_back((int)r.u64r());
  std::vector<uint32_t> kinds(n);
  r.u32arr(kinds.data(), n);
  c.constants_sigmas.assign(n_cs, std::vector<u64>(n));
  for (auto& p : c.constants_sigmas) r.u64arr(p.data(), n);
  c.rows.resize(n);
  for (size_t i = 0; i < n; i++) {
    c.rows[i].kind = (GateKind)kinds[i];
    c.rows[i].constants[0] = c.constants_sigmas[c.num_selectors][i];
    c.rows[i].constants[1] = c.constants_sigmas[c.num_selectors + 1][i];
  }
  c.k_is.resize(c.cfg.num_routed_wires);
  r.u64arr(c.k_is.data(), c.k_is.size());
  auto to_target = [&](uint32_t idx) -> Target {
    if (idx >= n * W) return Target{-1, (int32_t)(idx - n * W)};
    return Target{(int32_t)(idx / W), (int32_t)(idx % W)};
  };
  std::vector<uint32_t> in(n_in);
  r.u32arr(in.data(), n_in);
  for (auto i : in) c.input_targets.push_back(to_target(i));
  c.rep.resize(c.num_targets());
  r.u32arr(c.rep.data(), c.rep.size());
  c.generators.resize(n_gen);
  std::vector<uint32_t> args;
  for (auto& g : c.generators) {
    g.kind = (GenKind)r.u64r(); g.c0 = r.u64r(); g.c1 = r.u64r(); g.aux = (int)r.u64r();
    size_t nd = r.u64r(), no = r.u64r();
    if (g.kind >= GEN_NUM_KINDS || nd > 4096 || no > 4096) throw std::invalid_argument("circuit blob: bad generator");
    args.resize(nd + no);
    r.u32arr(args.data(), nd + no);
    for (size_t i = 0; i < nd; i++) g.deps.push_back(to_target(args[i]));
    for (size_t i = 0; i < no; i++) g.outs.push_back(to_target(args[nd + i]));
  }
  return c;
}
}  // namespace p25
