#include "circuit_io.h"
#include <algorithm>
#include <stdexcept>
#include <string>
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
// Inputs/outputs each generator body reads and writes (kernels_witgen.hip indexes them by kind).
bool generator_shape_ok(GenKind k, size_t nd, size_t no, int aux) {
  switch (k) {
    case GEN_CONSTANT: case GEN_RANDOM: return nd == 0 && no == 1;
    case GEN_ARITHMETIC: return nd == 3 && no == 1;
    case GEN_MUL_EXT: case GEN_QUOTIENT_EXT: return nd == 4 && no == 2;
    case GEN_BASE_SPLIT: return nd == 1 && no >= 1 && no <= 64;
    case GEN_WIRE_SPLIT: return nd == 1 && no >= 1 && no <= 2;
    case GEN_BASE_SUM: return nd >= 1 && nd <= 64 && no == 1;
    case GEN_LOW_HIGH: return nd == 1 && no == 2 && aux >= 1 && aux <= 63;
    case GEN_EXPONENTIATION: return nd >= 2 && nd <= 128 && no == nd;
    case GEN_POSEIDON2: case GEN_POSEIDON: return nd == 13 && no == 122;
    case GEN_ARITH_EXT: return nd == 6 && no == 2;
    case GEN_U32_ARITHMETIC: return nd == 3 && no == 35;
    case GEN_U32_INTERLEAVE: return nd == 1 && no == 33;
    case GEN_U32_UNINTERLEAVE: return nd == 1 && no == 66;
    case GEN_RANDOM_ACCESS: return nd == 1 + RA_VEC && no == 1 + RA_BITS;
    case GEN_REDUCING: return nd == 4 + RED_COEFFS && no == 2 * RED_COEFFS;
    case GEN_REDUCING_EXT: return nd == 4 + 2 * REDX_COEFFS && no == 2 * REDX_COEFFS;
    case GEN_COSET_INTERP: return nd == (size_t)CI_W_VALUE && no == 4 + 4 * CI_INTER;
    case GEN_POSEIDON_MDS: return nd == 24 && no == 24;
    default: return false;
  }
}
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
  h[20] = c.cfg.fri_arity_bits; h[21] = c.cfg.fri_final_poly_bits; h[22] = c.public_inputs.size();
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
  if (!c.public_inputs.empty()) {  // field 10 (absent without public inputs: earlier blobs stay valid)
    std::vector<uint32_t> pi(c.public_inputs.size());
    for (size_t i = 0; i < pi.size(); i++) pi[i] = (uint32_t)c.target_index(c.public_inputs[i]);
    w.u32arr(pi.data(), pi.size());
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
  const size_t n_pi = h[22];
  // Everything below is indexed by these fields, on the host and in the kernels: validate before use.
  auto bad = [](const char* what) { throw std::invalid_argument(std::string("circuit blob: ") + what); };
  for (int i = 0; i < 23; i++)
    if (i != 15 && h[i] > ((u64)1 << 31)) bad("header field out of range");
  if (n_pi > MAX_PUBLIC_INPUTS) bad("too many public inputs");
  if (n_pi && c.pi_row < 0) bad("public inputs without a PublicInputGate row");
  if (h[15] != (u64)-1 && h[15] >= ((u64)1 << 31)) bad("bad public-input row");
  if (c.degree_bits < 1 || c.degree_bits > 22) bad("degree_bits must be in 1..22");
  if (c.cfg.num_wires < 1 || c.cfg.num_wires > 1024 || c.cfg.num_routed_wires < 1 ||
      c.cfg.num_routed_wires > c.cfg.num_wires || c.cfg.num_routed_wires > MAX_ROUTED)
    bad("bad wire counts");
  if (c.cfg.num_challenges != 2) bad("num_challenges must be 2");
  if (c.cfg.rate_bits < 0 || c.cfg.rate_bits > 3 || c.cfg.max_quotient_degree_factor != (1 << c.cfg.rate_bits))
    bad("rate_bits must be in 0..3 with max_quotient_degree_factor = 2^rate_bits");
  if (c.cfg.cap_height < 0 || c.cfg.cap_height > 16 || c.cfg.cap_height > c.degree_bits + c.cfg.rate_bits)
    bad("cap_height exceeds the LDE size");
  if (c.cfg.num_query_rounds < 1 || c.cfg.num_query_rounds > 64) bad("num_query_rounds must be in 1..64");
  if (c.cfg.proof_of_work_bits < 0 || c.cfg.proof_of_work_bits > 32) bad("proof_of_work_bits must be in 0..32");
  if (ng < 1 || ng > G_NUM_KINDS || ng > 16) bad("bad gate count");
  if (c.num_selectors < 1 || (size_t)c.num_selectors > ng) bad("bad selector count");
  if (c.cfg.num_constants < 2 || c.cfg.num_constants > 64) bad("num_constants must be in 2..64");
  if (n_cs != (size_t)c.num_selectors + c.cfg.num_constants + c.cfg.num_routed_wires)
    bad("constants_sigmas count != selectors + constants + routed wires");
  if (c.num_partial_products !=
      (c.cfg.num_routed_wires + c.cfg.max_quotient_degree_factor - 1) / c.cfg.max_quotient_degree_factor - 1)
    bad("num_partial_products does not match ceil(num_routed / max_quotient_degree_factor) - 1");
  if (c.num_gate_constraints < 0 || c.num_gate_constraints > ALPHA_POWS) bad("too many gate constraints");
  // the permutation-argument kernels keep one value per chunk in registers and the quotient kernel folds with alpha
  // powers up to NC * (2 + NP): limits of the device code, not of the format
  if (c.num_partial_products + 1 > MAX_CHUNKS) bad("more partial-product chunks than the kernels hold (MAX_CHUNKS)");
  if (c.cfg.num_challenges * (2 + c.num_partial_products) >= ALPHA_POWS) bad("vanishing-polynomial terms exceed the alpha-power table");
  if (n_arity > 8) bad("too many FRI layers");
  if (c.pi_row != -1 && (c.pi_row < 0 || (size_t)c.pi_row >= ((size_t)1 << c.degree_bits))) bad("bad public-input row");
  {
    const u64 wires_total = ((u64)c.cfg.num_wires) << c.degree_bits;
    if (wires_total + h[16] >= ((u64)1 << 31)) bad("too many targets");
    // sizes the blob must at least carry (guards the allocations below against a forged header)
    const u64 need = 8 * ((n_cs << c.degree_bits) + (u64)c.cfg.num_routed_wires) + 4 * (((u64)1 << c.degree_bits) + n_in + wires_total + h[16]) + 48 * (u64)n_gen;
    if (need > len) bad("truncated");
  }
  const size_t n = c.degree();
  const int W = c.cfg.num_wires;
  c.groups.assign(c.num_selectors, {0, 0});
  for (size_t i = 0; i < ng; i++) {
    u64 k = r.u64r(), s = r.u64r(), gs = r.u64r(), ge = r.u64r();
    if (k >= G_NUM_KINDS || s >= (u64)c.num_selectors || gs > i || ge <= i || ge > ng) bad("bad gate table entry");
    for (GateKind seen : c.gates)
      if (seen == (GateKind)k) bad("duplicate gate kind");
    c.gates.push_back((GateKind)k); c.selector_index.push_back((int)s); c.groups[s] = {(int)gs, (int)ge};
  }
  {
    int max_nc = 0;
    for (GateKind k : c.gates) max_nc = std::max(max_nc, gate_info(k).num_constraints);
    if (c.num_gate_constraints != max_nc) bad("num_gate_constraints does not match the gate set");
    for (GateKind k : c.gates)
      if (gate_info(k).num_constants > c.cfg.num_constants) bad("a gate needs more constants than the circuit has");
    // A quotient of degree factor 2^rate_bits only exists if every FILTERED gate constraint fits in it: upstream
    // selectors.rs groups gates so that (gates in the group) + degree <= max_quotient_degree_factor + 1 (one more
    // when there is a single group and hence no "unused" selector value).
    for (size_t i = 0; i < c.gates.size(); i++) {
      const auto& gr = c.groups[c.selector_index[i]];
      if ((size_t)gr.first > i || (size_t)gr.second <= i) bad("gate outside its selector group");
      if ((gr.second - gr.first) + gate_info(c.gates[i]).degree > c.cfg.max_quotient_degree_factor + 1 + (c.num_selectors == 1 ? 1 : 0))
        bad("a filtered gate constraint exceeds the quotient degree (rate_bits too small for this gate set)");
    }
    // the evaluators index a row's wires by fixed column numbers
    static const int MIN_WIRES[G_NUM_KINDS] = {0, 2, 4, 1 + BASE_SUM_LIMBS, 6 + 96, 6 + 128, 80, 78, 2 + 2 * EXP_POWER_BITS,
                                               18 + 96, 135, 80, 135, RA_ROUTED + RA_BITS * RA_COPIES,
                                               4 + 3 * RED_COEFFS, 4 + 4 * REDX_COEFFS, CI_WIRES, 48};
    for (GateKind k : c.gates)
      if (MIN_WIRES[k] > c.cfg.num_wires) bad("a gate needs more wires than the circuit has");
  }
  {
    int bits = c.degree_bits + c.cfg.rate_bits, deg = c.degree_bits;
    for (size_t i = 0; i < n_arity; i++) {
      u64 a = r.u64r();
      if (a < 1 || a > 8) bad("FRI arity bits must be in 1..8");
      bits -= (int)a;
      deg -= (int)a;
      if (deg < 0 || bits < c.cfg.cap_height) bad("FRI layer smaller than the Merkle cap");
      c.fri_reduction_arity_bits.push_back((int)a);
    }
  }
  std::vector<uint32_t> kinds(n);
  r.u32arr(kinds.data(), n);
  {
    bool present[G_NUM_KINDS] = {false};
    for (GateKind k : c.gates) present[k] = true;
    for (size_t i = 0; i < n; i++)
      if (kinds[i] >= G_NUM_KINDS || !present[kinds[i]]) bad("row with a gate kind that is not in the gate table");
  }
  c.constants_sigmas.assign(n_cs, std::vector<u64>(n));
  for (auto& p : c.constants_sigmas) r.u64arr(p.data(), n);
  c.rows.resize(n);
  for (size_t i = 0; i < n; i++) {
    c.rows[i].kind = (GateKind)kinds[i];
    c.rows[i].constants[0] = c.constants_sigmas[c.num_selectors][i];
    c.rows[i].constants[1] = c.constants_sigmas[c.num_selectors + 1][i];
  }
  for (auto& p : c.constants_sigmas)
    for (u64 v : p)
      if (v >= gl::P) bad("non-canonical field element in constants_sigmas");
  c.k_is.resize(c.cfg.num_routed_wires);
  r.u64arr(c.k_is.data(), c.k_is.size());
  for (u64 v : c.k_is)
    if (v >= gl::P) bad("non-canonical k_i");
  const size_t NT = c.num_targets();
  auto to_target = [&](uint32_t idx) -> Target {
    if (idx >= n * W) return Target{-1, (int32_t)(idx - n * W)};
    return Target{(int32_t)(idx / W), (int32_t)(idx % W)};
  };
  std::vector<uint32_t> in(n_in);
  r.u32arr(in.data(), n_in);
  for (auto i : in) {
    if (i >= NT) bad("input target out of range");
    c.input_targets.push_back(to_target(i));
  }
  c.rep.resize(NT);
  r.u32arr(c.rep.data(), c.rep.size());
  for (uint32_t v : c.rep)
    if (v >= NT) bad("representative index out of range");
  c.generators.resize(n_gen);
  std::vector<uint32_t> args;
  for (auto& g : c.generators) {
    g.kind = (GenKind)r.u64r(); g.c0 = r.u64r(); g.c1 = r.u64r(); g.aux = (int)r.u64r();
    size_t nd = r.u64r(), no = r.u64r();
    if (g.kind >= GEN_NUM_KINDS || nd > 4096 || no > 4096) bad("bad generator");
    if (!generator_shape_ok(g.kind, nd, no, g.aux)) bad("generator with the wrong number of inputs/outputs for its kind");
    if (g.c0 >= gl::P || g.c1 >= gl::P) bad("non-canonical generator constant");
    args.resize(nd + no);
    r.u32arr(args.data(), nd + no);
    for (uint32_t a : args)
      if (a >= NT) bad("generator target out of range");
    for (size_t i = 0; i < nd; i++) g.deps.push_back(to_target(args[i]));
    for (size_t i = 0; i < no; i++) g.outs.push_back(to_target(args[nd + i]));
  }
  if (n_pi) {
    std::vector<uint32_t> pi(n_pi);
    r.u32arr(pi.data(), n_pi);
    for (uint32_t i : pi) {
      if (i >= NT) bad("public-input target out of range");
      c.public_inputs.push_back(to_target(i));
    }
  }
  if (r.off != len) bad("trailing bytes");
  return c;
}
}  // namespace p25
