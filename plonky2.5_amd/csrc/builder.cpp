// See builder.h.  Restates the behaviour of upstream plonky2's CircuitBuilder (plonky2 @ 3de92d9,
// circuit_builder.rs, gadgets/{arithmetic,arithmetic_extension,split_join,split_base,range_check,
// select}.rs, plonk/permutation_argument.rs, gates/selectors.rs) for the subset reachable from
// /root/reference/src/p3, and the reference's gadget layers cited per function.
#include "builder.h"
#include <algorithm>
#include <numeric>
#include <stdexcept>
#include <string.h>

namespace p25 {

// id() of the CosetInterpolationGate = format!("{self:?}<D={D}>") spells out its 16 barycentric weights (x_i / 16 on the
// subgroup of order 16): built once
static const std::string& coset_interp_id() {
  static const std::string id = [] {
    std::string s = "CosetInterpolationGate { subgroup_bits: 4, degree: 6, barycentric_weights: [";
    const u64 g = gl::root_of_unity(4), inv16 = gl::inv(16);
    u64 x = 1;
    for (int i = 0; i < CI_POINTS; i++) {
      s += (i ? ", " : "") + std::to_string(gl::mul(x, inv16));
      x = gl::mul(x, g);
    }
    return s + "], _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>";
  }();
  return id;
}
static const GateInfo GATE_INFOS[G_NUM_KINDS] = {
    {"NoopGate", 0, 0, 0, 1},
    {"ConstantGate { num_consts: 2 }", 1, 2, 2, 1},
    {"PublicInputGate", 1, 0, 4, 1},
    {"BaseSumGate { num_limbs: 63 } + Base: 2", 2, 0, 1 + BASE_SUM_LIMBS, 1},
    // id() = format!("{self:?}") of the reference structs (interleave_u32.rs:98-100 etc.)
    {"U32InterleaveGate { num_ops: 3 }", 2, 0, 3 * 34, 3},
    {"UninterleaveToU32Gate { num_ops: 2 }", 2, 0, 2 * 67, 2},
    {"ArithmeticGate { num_ops: 20 }", 3, 2, 20, 20},
    {"MulExtensionGate { num_ops: 13 }", 3, 1, 26, 13},
    {"ExponentiationGate { num_power_bits: 66, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>", 4, 0, EXP_POWER_BITS + 1, 1},
    {"U32ArithmeticGate { num_ops: 3, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }", 4, 0, 3 * 36, 3},
    {"Poseidon2Gate { _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<WIDTH=12>", 7, 0, 123, 1},
    {"ArithmeticExtensionGate { num_ops: 10 }", 3, 2, 20, 10},
    {"PoseidonGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>", 7, 0, 123, 1},
    {"RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2, _phantom: PhantomData<plonky2_field::goldilocks_field::GoldilocksField> }<D=2>",
     RA_BITS + 1, RA_EXTRA_CONSTS, RA_COPIES * (RA_BITS + 2) + RA_EXTRA_CONSTS, RA_COPIES},
    {"ReducingGate { num_coeffs: 43 }", 2, 0, 2 * RED_COEFFS, 1},
    {"ReducingExtensionGate { num_coeffs: 32 }", 2, 0, 2 * REDX_COEFFS, 1},
    {nullptr /* coset_interp_id() */, CI_DEGREE, 0, 2 * (2 + 2 * CI_INTER), 1},
    {"PoseidonMdsGate(PhantomData<plonky2_field::goldilocks_field::GoldilocksField>)<WIDTH=12>", 1, 0, 24, 1},
};
const GateInfo& gate_info(GateKind k) {
  if (k == G_COSET_INTERP) {
    static const GateInfo ci = {coset_interp_id().c_str(), GATE_INFOS[k].degree, GATE_INFOS[k].num_constants,
                                GATE_INFOS[k].num_constraints, GATE_INFOS[k].num_ops};
    return ci;
  }
  return GATE_INFOS[k];
}

// ---------------------------------------------------------------- core
Target CircuitBuilder::add_virtual_target() { return Target{-1, virtual_index_++}; }
std::vector<Target> CircuitBuilder::add_virtual_targets(int n) {
  std::vector<Target> v;
  for (int i = 0; i < n; i++) v.push_back(add_virtual_target());
  return v;
}
Target CircuitBuilder::constant(u64 c) {
  if (c >= gl::P) throw std::invalid_argument("non-canonical constant");
  auto it = constants_to_targets_.find(c);
  if (it != constants_to_targets_.end()) return it->second;
  Target t = add_virtual_target();
  constants_to_targets_[c] = t;
  targets_to_constants_[t] = c;
  return t;
}
std::optional<u64> CircuitBuilder::target_as_constant(Target t) const {
  auto it = targets_to_constants_.find(t);
  if (it == targets_to_constants_.end()) return std::nullopt;
  return it->second;
}
void CircuitBuilder::connect(Target x, Target y) {
  auto routable = [&](Target t) { return t.is_virtual() || t.col < config.num_routed_wires; };
  if (!routable(x) || !routable(y)) throw std::logic_error("connect: target is not routable");
  copy_constraints_.push_back({x, y});
}
int CircuitBuilder::add_gate(GateKind k, u64 c0, u64 c1) {
  int row = (int)rows_.size();
  if (k == G_CONSTANT)
    for (int i = 0; i < config.num_constants; i++) constant_generators_.push_back({row, i, i});
  rows_.push_back(GateInstance{k, {c0, c1}});
  return row;
}
// upstream CircuitBuilder::find_slot: one open row per (gate type, params)
std::pair<int, int> CircuitBuilder::find_slot(GateKind k, int n_params, u64 p0, u64 p1) {
  auto key = std::make_tuple((int)k, n_params > 0 ? p0 : 0, n_params > 1 ? p1 : 0, n_params);
  const int num_ops = gate_info(k).num_ops;
  auto it = current_slots_.find(key);
  int row, slot;
  if (it != current_slots_.end()) {
    row = it->second.first;
    slot = it->second.second;
  } else {
    row = add_gate(k, n_params > 0 ? p0 : 0, n_params > 1 ? p1 : 0);
    slot = 0;
  }
  if (slot == num_ops - 1)
    current_slots_.erase(key);
  else
    current_slots_[key] = {row, slot + 1};
  return {row, slot};
}

// ---------------------------------------------------------------- base arithmetic
Target CircuitBuilder::arithmetic(u64 c0, u64 c1, Target m0, Target m1, Target addend) {
  // arithmetic_special_cases
  Target z = zero();
  auto k0 = target_as_constant(m0), k1 = target_as_constant(m1), ka = target_as_constant(addend);
  bool first_zero = c0 == 0 || m0 == z || m1 == z;
  bool second_zero = c1 == 0 || addend == z;
  std::optional<u64> first_const, second_const;
  if (first_zero)
    first_const = 0;
  else if (k0 && k1)
    first_const = gl::mul(gl::mul(*k0, *k1), c0);
  if (second_zero)
    second_const = 0;
  else if (ka)
    second_const = gl::mul(*ka, c1);
  if (first_const && second_const) return constant(gl::add(*first_const, *second_const));
  if (first_zero && c1 == 1) return addend;
  if (second_zero) {
    if (k0 && gl::mul(*k0, c0) == 1) return m1;
    if (k1 && gl::mul(*k1, c0) == 1) return m0;
  }
  ArithKey key{c0, c1, m0, m1, addend};
  auto it = base_arithmetic_results_.find(key);
  if (it != base_arithmetic_results_.end()) return it->second;
  auto [row, i] = find_slot(G_ARITHMETIC, 2, c0, c1);
  connect(m0, wire(row, 4 * i));
  connect(m1, wire(row, 4 * i + 1));
  connect(addend, wire(row, 4 * i + 2));
  Target out = wire(row, 4 * i + 3);
  base_arithmetic_results_[key] = out;
  return out;
}
Target CircuitBuilder::exp_power_of_2(Target base, int power_log) {
  Target p = base;
  for (int i = 0; i < power_log; i++) p = square(p);
  return p;
}
Target CircuitBuilder::exp_from_bits(Target base, const std::vector<BoolTarget>& bits_in) {
  BoolTarget f = _false();
  std::vector<BoolTarget> bits(bits_in);
  while ((int)bits.size() < EXP_POWER_BITS) bits.push_back(f);
  int row = add_gate(G_EXPONENTIATION);
  connect(base, wire(row, 0));
  for (int i = 0; i < EXP_POWER_BITS; i++) connect(bits[i], wire(row, 1 + i));
  return wire(row, 1 + EXP_POWER_BITS);
}
Target CircuitBuilder::exp_u64(Target base, u64 exponent) {
  std::vector<BoolTarget> bits;
  while (exponent) {
    bits.push_back(constant_bool(exponent & 1));
    exponent >>= 1;
  }
  return exp_from_bits(base, bits);
}
Target CircuitBuilder::exp(Target base, Target exponent, int num_bits) {
  auto bits = split_le(exponent, num_bits);
  return exp_from_bits(base, bits);
}
Target CircuitBuilder::select(BoolTarget b, Target x, Target y) {
  Target tmp = mul_sub(b, y, y);
  return mul_sub(b, x, tmp);
}

// upstream gadgets/arithmetic_extension.rs `arithmetic_extension`: c0 * m0 * m1 + c1 * addend in F_p^2.  Special
// cases first (arithmetic_extension_special_cases), then the memo, then a MulExtensionGate slot when the addend is
// the zero constant and an ArithmeticExtensionGate slot otherwise.
std::optional<gl::E2> CircuitBuilder::target_as_constant_ext(Ext e) const {
  auto x = target_as_constant(e[0]), y = target_as_constant(e[1]);
  if (x && y) return gl::E2{*x, *y};
  return std::nullopt;
}
Ext CircuitBuilder::arithmetic_extension(u64 c0, u64 c1, Ext m0, Ext m1, Ext addend) {
  const Ext zext = zero_extension();
  auto k0 = target_as_constant_ext(m0), k1 = target_as_constant_ext(m1), ka = target_as_constant_ext(addend);
  const bool first_zero = c0 == 0 || m0 == zext || m1 == zext;
  const bool second_zero = c1 == 0 || addend == zext;
  std::optional<gl::E2> first_const, second_const;
  if (first_zero)
    first_const = gl::E2{0, 0};
  else if (k0 && k1)
    first_const = gl::mul(gl::mul(*k0, *k1), c0);
  if (second_zero)
    second_const = gl::E2{0, 0};
  else if (ka)
    second_const = gl::mul(*ka, c1);
  if (first_const && second_const) return constant_extension(gl::add(*first_const, *second_const));
  if (first_zero && c1 == 1) return addend;
  if (second_zero) {
    auto is_one = [](gl::E2 x) { return x.a == 1 && x.b == 0; };
    if (k0 && is_one(gl::mul(*k0, c0))) return m1;
    if (k1 && is_one(gl::mul(*k1, c0))) return m0;
  }
  ExtArithKey key{c0, c1, {m0[0], m0[1], m1[0], m1[1], addend[0], addend[1]}};
  auto it = ext_arithmetic_results_.find(key);
  if (it != ext_arithmetic_results_.end()) return it->second;
  Ext out;
  if (ka && ka->a == 0 && ka->b == 0) {  // addend is the zero constant: MulExtensionGate
    auto [row, i] = find_slot(G_MUL_EXT, 1, c0, 0);
    for (int d = 0; d < 2; d++) {
      connect(m0[d], wire(row, 6 * i + d));
      connect(m1[d], wire(row, 6 * i + 2 + d));
    }
    out = {wire(row, 6 * i + 4), wire(row, 6 * i + 5)};
  } else {
    auto [row, i] = find_slot(G_ARITH_EXT, 2, c0, c1);
    for (int d = 0; d < 2; d++) {
      connect(m0[d], wire(row, 8 * i + d));
      connect(m1[d], wire(row, 8 * i + 2 + d));
      connect(addend[d], wire(row, 8 * i + 4 + d));
    }
    out = {wire(row, 8 * i + 6), wire(row, 8 * i + 7)};
  }
  ext_arithmetic_results_[key] = out;
  return out;
}
Ext CircuitBuilder::mul_extension(Ext a, Ext b) { return arithmetic_extension(1, 0, a, b, zero_extension()); }
Ext CircuitBuilder::add_many_extension(const std::vector<Ext>& terms) {
  Ext sum = zero_extension();
  for (const Ext& t : terms) sum = add_extension(sum, t);
  return sum;
}
Ext CircuitBuilder::mul_many_extension(const std::vector<Ext>& terms) {
  Ext prod = one_extension();
  for (const Ext& t : terms) prod = mul_extension(prod, t);
  return prod;
}
Ext CircuitBuilder::exp_u64_extension(Ext base, u64 exponent) {
  if (exponent == 0) return one_extension();
  if (exponent == 1) return base;
  if (exponent == 2) return square_extension(base);
  Ext current = base, product = one_extension();
  int bits = 64 - __builtin_clzll(exponent);
  for (int j = 0; j < bits; j++) {
    if (j != 0) current = square_extension(current);
    if ((exponent >> j) & 1) product = mul_extension(product, current);
  }
  return product;
}
Ext CircuitBuilder::exp_power_of_2_extension(Ext base, int power_log) {
  for (int i = 0; i < power_log; i++) base = square_extension(base);
  return base;
}
// x / y + z with a virtual inverse and QuotientGeneratorExtension (y * inv == 1 enforced)
Ext CircuitBuilder::div_add_extension(Ext x, Ext y, Ext z) {
  Ext inv = add_virtual_extension_target();
  Ext one_e = one_extension();
  Generator g;
  g.kind = GEN_QUOTIENT_EXT;
  g.deps = {one_e[0], one_e[1], y[0], y[1]};
  g.outs = {inv[0], inv[1]};
  add_generator(g);
  Ext y_inv = mul_extension(y, inv);
  connect_extension(y_inv, one_e);
  return mul_add_extension(x, inv, z);
}
// select(b, x, y) = b*x + (1-b)*y  (upstream gadgets/select.rs: tmp = b*y - y; b*x - tmp)
Ext CircuitBuilder::select_ext(BoolTarget b, Ext x, Ext y) {
  Ext be = convert_to_ext(b);
  Ext tmp = mul_sub_extension(be, y, y);
  return mul_sub_extension(be, x, tmp);
}

// inverse(x) = inverse_extension([x, 0]).0[0] = div_add_extension(one, [x,0], zero).0[0]
Target CircuitBuilder::inverse(Target x) {
  Target z = zero();
  Ext y = {x, z};
  Ext inv = {add_virtual_target(), add_virtual_target()};
  Ext one_e = {one(), zero()};
  Generator g;
  g.kind = GEN_QUOTIENT_EXT;
  g.deps = {one_e[0], one_e[1], y[0], y[1]};
  g.outs = {inv[0], inv[1]};
  add_generator(g);
  Ext y_inv = mul_extension(y, inv);
  connect(y_inv[0], one_e[0]);
  connect(y_inv[1], one_e[1]);
  // mul_add_extension(one, inv, zero) hits the "multiplicand is one, addend zero" special case
  return inv[0];
}

// ---------------------------------------------------------------- split / join
std::vector<BoolTarget> CircuitBuilder::split_le(Target integer, int num_bits) {
  if (num_bits == 0) return {};
  const int L = BASE_SUM_LIMBS;
  int k = (num_bits + L - 1) / L;
  std::vector<int> gates;
  for (int i = 0; i < k; i++) gates.push_back(add_gate(G_BASE_SUM));
  std::vector<BoolTarget> bits;
  for (int g : gates)
    for (int l = 0; l < L; l++) bits.push_back(wire(g, 1 + l));
  for (size_t i = num_bits; i < bits.size(); i++) assert_zero(bits[i]);
  bits.resize(num_bits);
  Target acc = zero();
  u64 base = gl::pow(2, L);
  for (int gi = k - 1; gi >= 0; gi--) acc = mul_const_add(base, acc, wire(gates[gi], 0));
  connect(acc, integer);
  Generator g;
  g.kind = GEN_WIRE_SPLIT;
  g.deps = {integer};
  for (int gt : gates) g.outs.push_back(wire(gt, 0));
  add_generator(g);
  return bits;
}
Target CircuitBuilder::le_sum(const std::vector<BoolTarget>& bits) {
  int num_bits = (int)bits.size();
  if (num_bits == 0) return zero();
  if (num_bits - 1 <= gate_info(G_ARITHMETIC).num_ops) {
    Target sum = bits[num_bits - 1];
    for (int i = num_bits - 2; i >= 0; i--) sum = mul_const_add(2, sum, bits[i]);
    return sum;
  }
  int row = add_gate(G_BASE_SUM);
  for (int i = 0; i < num_bits; i++) connect(bits[i], wire(row, 1 + i));
  for (int l = num_bits; l < BASE_SUM_LIMBS; l++) assert_zero(wire(row, 1 + l));
  Generator g;
  g.kind = GEN_BASE_SUM;
  g.deps = bits;
  g.outs = {wire(row, 0)};
  add_generator(g);
  return wire(row, 0);
}
std::pair<Target, Target> CircuitBuilder::split_low_high(Target x, int n_log, int num_bits) {
  Target low = add_virtual_target(), high = add_virtual_target();
  Generator g;
  g.kind = GEN_LOW_HIGH;
  g.aux = n_log;
  g.deps = {x};
  g.outs = {low, high};
  add_generator(g);
  range_check(low, n_log);
  range_check(high, num_bits - n_log);
  Target pow2 = constant((u64)1 << n_log);
  Target comb = mul_add(high, pow2, low);
  connect(x, comb);
  return {low, high};
}

// ---------------------------------------------------------------- u32 gadgets (reference)
// src/common/u32/gadgets/arithmetic_u32.rs:120-178
std::pair<Target, Target> CircuitBuilder::mul_add_u32(Target x, Target y, Target z) {
  auto kx = target_as_constant(x), ky = target_as_constant(y), kz = target_as_constant(z);
  if (kx && ky && kz) {
    u64 sum = gl::add(gl::mul(*kx, *ky), *kz);
    return {constant_u32((uint32_t)sum), constant_u32((uint32_t)(sum >> 32))};
  }
  auto [row, i] = find_slot(G_U32_ARITHMETIC, 0, 0, 0);
  connect(wire(row, 6 * i), x);
  connect(wire(row, 6 * i + 1), y);
  connect(wire(row, 6 * i + 2), z);
  return {wire(row, 6 * i + 3), wire(row, 6 * i + 4)};
}
// src/common/u32/gadgets/interleaved_u32.rs:89-111
Target CircuitBuilder::interleave_u32(Target x) {
  auto [row, i] = find_slot(G_U32_INTERLEAVE, 0, 0, 0);
  connect(wire(row, 2 * i), x);
  return wire(row, 2 * i + 1);
}
std::pair<Target, Target> CircuitBuilder::uninterleave_to_u32(Target x) {
  auto [row, i] = find_slot(G_U32_UNINTERLEAVE, 0, 0, 0);
  connect(wire(row, 3 * i), x);
  return {wire(row, 3 * i + 1), wire(row, 3 * i + 2)};
}
// interleaved_u32.rs:193-224
std::pair<Target, Target> CircuitBuilder::and_xor_u32_to_u32(Target x, Target y) {
  Target xi = interleave_u32(x);
  Target yi = interleave_u32(y);
  Target sum = add(xi, yi);
  return uninterleave_to_u32(sum);
}
std::array<Target, 2> CircuitBuilder::and_u64(std::array<Target, 2> x, std::array<Target, 2> y) {
  Target a = and_u32(x[0], y[0]);
  Target b = and_u32(x[1], y[1]);
  return {a, b};
}
std::array<Target, 2> CircuitBuilder::xor_u64(std::array<Target, 2> x, std::array<Target, 2> y) {
  Target a = xor_u32(x[0], y[0]);
  Target b = xor_u32(x[1], y[1]);
  return {a, b};
}
// interleaved_u32.rs:290-312
std::array<Target, 2> CircuitBuilder::lsh_u64(std::array<Target, 2> x, int n) {
  if (n == 0) return x;
  Target lo = x[0], hi = x[1];
  if (n < 32) {
    Target p2 = constant_u32(1u << (n % 32));
    auto [lo0, hi0] = mul_u32(lo, p2);
    auto [lo1, hi1] = mul_u32(hi, p2);
    (void)hi1;
    Target h = add_u32(hi0, lo1).first;
    return {lo0, h};
  }
  Target p2 = constant_u32(1u << (n % 32));
  auto [lo0, hi0] = mul_u32(lo, p2);
  (void)hi0;
  return {zero(), lo0};
}
// interleaved_u32.rs:314-335
std::array<Target, 2> CircuitBuilder::rsh_u64(std::array<Target, 2> x, int n) {
  if (n == 0) return x;
  Target lo = x[0], hi = x[1];
  if (n < 32) {
    Target p2 = constant_u32(1u << (32 - (n % 32)));
    auto [lo0, hi0] = mul_u32(lo, p2);
    (void)lo0;
    auto [lo1, hi1] = mul_u32(hi, p2);
    Target l = add_u32(lo1, hi0).first;
    return {l, hi1};
  }
  // (n % 32 == 0 would overflow the u32 shift in the reference too; not reachable from src/p3)
  Target p2 = constant_u32(1u << (32 - (n % 32)));
  auto [lo1, hi1] = mul_u32(hi, p2);
  (void)lo1;
  return {hi1, zero()};
}

// src/common/poseidon2/poseidon2.rs:585-609 (Poseidon2Hash::permute_targets)
std::array<Target, 12> CircuitBuilder::poseidon2_permute_targets(const std::array<Target, 12>& in) {
  int row = add_gate(G_POSEIDON2);
  Target swap = zero();
  connect(swap, wire(row, 24));
  for (int i = 0; i < 12; i++) connect(in[i], wire(row, i));
  std::array<Target, 12> out;
  for (int i = 0; i < 12; i++) out[i] = wire(row, 12 + i);
  return out;
}

// upstream hash/poseidon.rs `PoseidonHash::permute_swapped` on a PoseidonGate row
std::array<Target, 12> CircuitBuilder::poseidon_permute_swapped(const std::array<Target, 12>& in, BoolTarget swap) {
  int row = add_gate(G_POSEIDON);
  connect(swap, wire(row, 24));
  for (int i = 0; i < 12; i++) connect(in[i], wire(row, i));
  std::array<Target, 12> out;
  for (int i = 0; i < 12; i++) out[i] = wire(row, 12 + i);
  return out;
}
// upstream hash/hashing.rs `hash_n_to_m_no_pad` in-circuit (overwrite mode, rate 8), first 4 outputs
std::array<Target, 4> CircuitBuilder::hash_n_to_hash_no_pad(const std::vector<Target>& inputs) {
  std::array<Target, 12> state;
  for (auto& t : state) t = zero();
  for (size_t off = 0; off < inputs.size(); off += 8) {
    for (size_t i = 0; i < 8 && off + i < inputs.size(); i++) state[i] = inputs[off + i];
    state = poseidon_permute(state);
  }
  return {state[0], state[1], state[2], state[3]};
}
std::array<Target, 4> CircuitBuilder::hash_or_noop(const std::vector<Target>& inputs) {
  if (inputs.size() <= 4) {
    std::array<Target, 4> h;
    for (int i = 0; i < 4; i++) h[i] = i < (int)inputs.size() ? inputs[i] : zero();
    return h;
  }
  return hash_n_to_hash_no_pad(inputs);
}

// ---------------------------------------------------------------- random access (upstream gadgets/random_access.rs)
Target CircuitBuilder::random_access(Target access_index, const std::vector<Target>& v) {
  if (v.size() == 1) return v[0];
  if ((int)v.size() != RA_VEC) throw std::invalid_argument("random_access: lists of 16 (or 1) elements");
  Target claimed = add_virtual_target();
  auto [row, copy] = find_slot(G_RANDOM_ACCESS, 0, 0, 0);
  const int base = (2 + RA_VEC) * copy;
  for (int i = 0; i < RA_VEC; i++) connect(v[i], wire(row, base + 2 + i));
  connect(access_index, wire(row, base));
  connect(claimed, wire(row, base + 1));
  return claimed;
}
Ext CircuitBuilder::random_access_extension(Target access_index, const std::vector<Ext>& v) {
  Ext r;
  for (int d = 0; d < 2; d++) {
    std::vector<Target> comp;
    for (const Ext& e : v) comp.push_back(e[d]);
    r[d] = random_access(access_index, comp);
  }
  return r;
}
std::array<Target, 4> CircuitBuilder::random_access_hash(Target access_index, const std::vector<std::array<Target, 4>>& v) {
  std::array<Target, 4> r;
  for (int i = 0; i < 4; i++) {
    std::vector<Target> comp;
    for (const auto& h : v) comp.push_back(h[i]);
    r[i] = random_access(access_index, comp);
  }
  return r;
}

Ext CircuitBuilder::interpolate_coset(Target coset_shift, const std::vector<Ext>& values, Ext evaluation_point) {
  if ((int)values.size() != CI_POINTS) throw std::invalid_argument("interpolate_coset: 16 values");
  const int row = add_gate(G_COSET_INTERP);
  connect(coset_shift, wire(row, 0));
  for (int i = 0; i < CI_POINTS; i++) connect_extension(values[i], Ext{wire(row, 1 + 2 * i), wire(row, 2 + 2 * i)});
  connect_extension(evaluation_point, Ext{wire(row, CI_W_POINT), wire(row, CI_W_POINT + 1)});
  return Ext{wire(row, CI_W_VALUE), wire(row, CI_W_VALUE + 1)};
}

std::array<Ext, 12> CircuitBuilder::poseidon_mds_layer(const std::array<Ext, 12>& state) {
  const int row = add_gate(G_POSEIDON_MDS);
  std::array<Ext, 12> out;
  for (int i = 0; i < 12; i++) {
    connect_extension(state[i], Ext{wire(row, 2 * i), wire(row, 2 * i + 1)});
    out[i] = Ext{wire(row, 24 + 2 * i), wire(row, 25 + 2 * i)};
  }
  return out;
}

// ---------------------------------------------------------------- build
std::vector<int> fri_reduction_arity_bits(const CircuitConfig& cfg, int degree_bits) {
  std::vector<int> r;
  int db = degree_bits;
  while (db > cfg.fri_final_poly_bits && db + cfg.rate_bits - cfg.fri_arity_bits >= cfg.cap_height) {
    r.push_back(cfg.fri_arity_bits);
    db -= cfg.fri_arity_bits;
  }
  return r;
}

namespace {
struct Dsu {
  std::vector<uint32_t> p;
  explicit Dsu(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
  uint32_t find(uint32_t x) {
    while (p[x] != x) {
      p[x] = p[p[x]];
      x = p[x];
    }
    return x;
  }
  void merge(uint32_t a, uint32_t b) {
    a = find(a);
    b = find(b);
    if (a != b) p[a] = b;
  }
};
}  // namespace

// The witness generators a gate row contributes (upstream `Gate::generators(row, local_constants)`): one per
// operation slot for the multi-op gates, one per row for the others, none for gates whose wires are set by explicit
// generators (ConstantGate, PublicInputGate) or by nothing (NoopGate).  Wire layouts as in the gate files cited in
// kernels_witgen.hip.  Shared by `build()` and by the reader of upstream's `CircuitData` bytes (circuit_bytes.cpp).
int gate_generator_ops(GateKind k) {
  switch (k) {
    case G_BASE_SUM:
    case G_EXPONENTIATION:
    case G_POSEIDON:
    case G_POSEIDON2:
      return 1;
    case G_ARITHMETIC:
    case G_MUL_EXT:
    case G_U32_ARITHMETIC:
    case G_U32_INTERLEAVE:
    case G_U32_UNINTERLEAVE:
    case G_ARITH_EXT:
    case G_RANDOM_ACCESS:
      return gate_info(k).num_ops;
    case G_REDUCING:
    case G_REDUCING_EXT:
    case G_COSET_INTERP:
    case G_POSEIDON_MDS:
      return 1;
    default:
      return 0;
  }
}
Generator gate_op_generator(GateKind kind, const u64 constants[2], int r, int i) {
  Generator g;
  switch (kind) {
    case G_BASE_SUM:
      g.kind = GEN_BASE_SPLIT;
      g.deps = {wire(r, 0)};
      for (int l = 0; l < BASE_SUM_LIMBS; l++) g.outs.push_back(wire(r, 1 + l));
      break;
    case G_ARITHMETIC:
      g.kind = GEN_ARITHMETIC;
      g.c0 = constants[0];
      g.c1 = constants[1];
      g.deps = {wire(r, 4 * i), wire(r, 4 * i + 1), wire(r, 4 * i + 2)};
      g.outs = {wire(r, 4 * i + 3)};
      break;
    case G_MUL_EXT:
      g.kind = GEN_MUL_EXT;
      g.c0 = constants[0];
      g.deps = {wire(r, 6 * i), wire(r, 6 * i + 1), wire(r, 6 * i + 2), wire(r, 6 * i + 3)};
      g.outs = {wire(r, 6 * i + 4), wire(r, 6 * i + 5)};
      break;
    case G_EXPONENTIATION:
      g.kind = GEN_EXPONENTIATION;
      for (int k = 0; k <= EXP_POWER_BITS; k++) g.deps.push_back(wire(r, k));  // base, bits
      for (int k = 0; k < EXP_POWER_BITS; k++) g.outs.push_back(wire(r, 2 + EXP_POWER_BITS + k));
      g.outs.push_back(wire(r, 1 + EXP_POWER_BITS));
      break;
    case G_U32_ARITHMETIC:
      g.kind = GEN_U32_ARITHMETIC;
      g.deps = {wire(r, 6 * i), wire(r, 6 * i + 1), wire(r, 6 * i + 2)};
      g.outs = {wire(r, 6 * i + 3), wire(r, 6 * i + 4), wire(r, 6 * i + 5)};
      for (int j = 0; j < 32; j++) g.outs.push_back(wire(r, 18 + 32 * i + j));
      break;
    case G_U32_INTERLEAVE:
      g.kind = GEN_U32_INTERLEAVE;
      g.deps = {wire(r, 2 * i)};
      for (int j = 0; j < 32; j++) g.outs.push_back(wire(r, 6 + 32 * i + j));
      g.outs.push_back(wire(r, 2 * i + 1));
      break;
    case G_U32_UNINTERLEAVE:
      g.kind = GEN_U32_UNINTERLEAVE;
      g.deps = {wire(r, 3 * i)};
      for (int j = 0; j < 64; j++) g.outs.push_back(wire(r, 6 + 64 * i + j));
      g.outs.push_back(wire(r, 3 * i + 1));
      g.outs.push_back(wire(r, 3 * i + 2));
      break;
    case G_ARITH_EXT:
      g.kind = GEN_ARITH_EXT;
      g.c0 = constants[0];
      g.c1 = constants[1];
      for (int k = 0; k < 6; k++) g.deps.push_back(wire(r, 8 * i + k));
      g.outs = {wire(r, 8 * i + 6), wire(r, 8 * i + 7)};
      break;
    case G_POSEIDON:
    case G_POSEIDON2:
      g.kind = kind == G_POSEIDON ? GEN_POSEIDON : GEN_POSEIDON2;
      for (int k = 0; k < 12; k++) g.deps.push_back(wire(r, k));
      g.deps.push_back(wire(r, 24));
      for (int k = 0; k < 4; k++) g.outs.push_back(wire(r, 25 + k));     // delta
      for (int k = 0; k < 106; k++) g.outs.push_back(wire(r, 29 + k));   // S-box inputs
      for (int k = 0; k < 12; k++) g.outs.push_back(wire(r, 12 + k));    // outputs
      break;
    case G_RANDOM_ACCESS: {  // upstream RandomAccessGenerator: the index and the list -> the element and the index bits
      g.kind = GEN_RANDOM_ACCESS;
      const int base = (2 + RA_VEC) * i;
      g.deps.push_back(wire(r, base));
      for (int k = 0; k < RA_VEC; k++) g.deps.push_back(wire(r, base + 2 + k));
      g.outs.push_back(wire(r, base + 1));
      for (int k = 0; k < RA_BITS; k++) g.outs.push_back(wire(r, RA_ROUTED + RA_BITS * i + k));
      break;
    }
    case G_REDUCING:
    case G_REDUCING_EXT: {  // upstream ReducingGenerator: alpha, old_acc, coefficients -> every accumulator
      const bool ext = kind == G_REDUCING_EXT;
      const int nco = ext ? REDX_COEFFS : RED_COEFFS, cw = ext ? 2 : 1, start_accs = 6 + nco * cw;
      g.kind = ext ? GEN_REDUCING_EXT : GEN_REDUCING;
      for (int k = 2; k < 6 + nco * cw; k++) g.deps.push_back(wire(r, k));        // alpha, old_acc, coefficients
      for (int k = 0; k < nco; k++) {
        const int w0 = k == nco - 1 ? 0 : start_accs + 2 * k;                      // the last accumulator is the output
        g.outs.push_back(wire(r, w0));
        g.outs.push_back(wire(r, w0 + 1));
      }
      break;
    }
    case G_POSEIDON_MDS:  // upstream PoseidonMdsGenerator
      g.kind = GEN_POSEIDON_MDS;
      for (int k = 0; k < 24; k++) g.deps.push_back(wire(r, k));
      for (int k = 0; k < 24; k++) g.outs.push_back(wire(r, 24 + k));
      break;
    case G_COSET_INTERP:  // upstream InterpolationGenerator
      g.kind = GEN_COSET_INTERP;
      for (int k = 0; k < CI_W_VALUE; k++) g.deps.push_back(wire(r, k));            // shift, 16 values, the point
      g.outs.push_back(wire(r, CI_W_SHIFTED));
      g.outs.push_back(wire(r, CI_W_SHIFTED + 1));
      for (int k = 0; k < CI_INTER; k++) {
        g.outs.push_back(wire(r, CI_W_INTER + 2 * k));                               // intermediate eval k
        g.outs.push_back(wire(r, CI_W_INTER + 2 * k + 1));
        g.outs.push_back(wire(r, CI_W_INTER + 2 * (CI_INTER + k)));                  // intermediate product k
        g.outs.push_back(wire(r, CI_W_INTER + 2 * (CI_INTER + k) + 1));
      }
      g.outs.push_back(wire(r, CI_W_VALUE));
      g.outs.push_back(wire(r, CI_W_VALUE + 1));
      break;
    default:
      throw std::logic_error("gate_op_generator: gate without generators");
  }
  return g;
}

Circuit CircuitBuilder::build() {
  Circuit c;
  c.cfg = config;
  const int W = config.num_wires, RW = config.num_routed_wires;

  // "Hash the public inputs, and route them to a PublicInputGate which will enforce that they hash to the expected
  // value" (upstream build()): hash_n_to_hash_no_pad of the registered targets on PoseidonGate rows -- of [] it is 4
  // zeros and costs no row; the gate's unused wires get RandomValueGenerators (upstream randomize_unused_pi_wires)
  c.public_inputs = public_inputs_;
  std::array<Target, 4> pi_hash = hash_n_to_hash_no_pad(public_inputs_);
  int pi_gate = add_gate(G_PUBLIC_INPUT);
  for (int i = 0; i < 4; i++) connect(pi_hash[i], wire(pi_gate, i));
  for (int w = 4; w < W; w++) {
    Generator g;
    g.kind = GEN_RANDOM;
    g.aux = w;
    g.outs = {wire(pi_gate, w)};
    add_generator(g);
  }
  c.pi_row = pi_gate;

  // constants: enough ConstantGates, then constants in increasing order -> constant generators
  while (constants_to_targets_.size() > constant_generators_.size()) add_gate(G_CONSTANT);
  {
    size_t i = 0;
    for (auto& [val, t] : constants_to_targets_) {  // std::map: sorted by canonical value
      const ConstGen& cg = constant_generators_[i++];
      rows_[cg.row].constants[cg.constant_index] = val;
      connect(wire(cg.row, cg.wire_index), t);
      Generator g;
      g.kind = GEN_CONSTANT;
      g.c0 = val;
      g.outs = {wire(cg.row, cg.wire_index)};
      add_generator(g);
    }
  }
  // pad to a power of two with NoopGates
  while (rows_.size() & (rows_.size() - 1)) add_gate(G_NOOP);
  const size_t n = rows_.size();
  int degree_bits = 0;
  while (((size_t)1 << degree_bits) < n) degree_bits++;
  c.degree_bits = degree_bits;
  c.rows = rows_;
  c.fri_reduction_arity_bits = fri_reduction_arity_bits(config, degree_bits);
  c.num_virtual_targets = virtual_index_;
  c.input_targets = input_targets;

  // gate set sorted by (degree, id)
  bool present[G_NUM_KINDS] = {false};
  for (auto& r : rows_) present[r.kind] = true;
  for (int k = 0; k < G_NUM_KINDS; k++)
    if (present[k]) c.gates.push_back((GateKind)k);
  std::sort(c.gates.begin(), c.gates.end(), [](GateKind a, GateKind b) {
    const GateInfo &x = gate_info(a), &y = gate_info(b);
    if (x.degree != y.degree) return x.degree < y.degree;
    return strcmp(x.id, y.id) < 0;
  });
  const int num_gates = (int)c.gates.size();
  int gate_index_of[G_NUM_KINDS];
  for (int i = 0; i < G_NUM_KINDS; i++) gate_index_of[i] = -1;
  for (int i = 0; i < num_gates; i++) gate_index_of[c.gates[i]] = i;

  // selector polynomials (upstream selectors.rs::selector_polynomials, max_degree = qdf + 1)
  const int max_degree = config.max_quotient_degree_factor + 1;
  const int max_gate_degree = gate_info(c.gates.back()).degree;
  std::vector<std::vector<u64>> selector_polys;
  if (max_gate_degree + num_gates - 1 <= max_degree) {
    c.groups = {{0, num_gates}};
    c.selector_index.assign(num_gates, 0);
    std::vector<u64> s(n);
    for (size_t j = 0; j < n; j++) s[j] = gate_index_of[rows_[j].kind];
    selector_polys.push_back(s);
  } else {
    if (max_gate_degree >= max_degree) throw std::logic_error("gate degree too high");
    int start = 0;
    while (start < num_gates) {
      int size = 0;
      while (start + size < num_gates && size + gate_info(c.gates[start + size]).degree < max_degree) size++;
      c.groups.push_back({start, start + size});
      start += size;
    }
    c.selector_index.resize(num_gates);
    for (int i = 0; i < num_gates; i++)
      for (size_t g = 0; g < c.groups.size(); g++)
        if (i >= c.groups[g].first && i < c.groups[g].second) c.selector_index[i] = (int)g;
    const u64 UNUSED = 0xFFFFFFFFull;  // UNUSED_SELECTOR = u32::MAX
    selector_polys.assign(c.groups.size(), std::vector<u64>(n));
    for (size_t j = 0; j < n; j++) {
      int gi = gate_index_of[rows_[j].kind];
      int gr = c.selector_index[gi];
      for (size_t g = 0; g < c.groups.size(); g++) selector_polys[g][j] = (int)g == gr ? (u64)gi : UNUSED;
    }
  }
  c.num_selectors = (int)selector_polys.size();
  c.num_gate_constraints = 0;
  for (GateKind k : c.gates) c.num_gate_constraints = std::max(c.num_gate_constraints, gate_info(k).num_constraints);
  c.num_partial_products = (RW + config.max_quotient_degree_factor - 1) / config.max_quotient_degree_factor - 1;

  // constant polynomials
  int max_constants = 0;
  for (GateKind k : c.gates) max_constants = std::max(max_constants, gate_info(k).num_constants);
  c.constants_sigmas = selector_polys;
  for (int ci = 0; ci < max_constants; ci++) {
    std::vector<u64> p(n);
    for (size_t j = 0; j < n; j++) p[j] = rows_[j].constants[ci];
    c.constants_sigmas.push_back(std::move(p));
  }

  // copy constraints -> partition -> sigmas
  c.k_is.resize(RW);
  {
    u64 x = 1;
    for (int i = 0; i < RW; i++) {
      c.k_is[i] = x;
      x = gl::mul(x, gl::GENERATOR);
    }
  }
  const size_t n_targets = n * W + (size_t)virtual_index_;
  auto tindex = [&](Target t) -> uint32_t {
    return (uint32_t)(t.is_virtual() ? n * W + t.col : (size_t)t.row * W + t.col);
  };
  Dsu dsu(n_targets);
  for (auto& cc : copy_constraints_) dsu.merge(tindex(cc.first), tindex(cc.second));
  c.rep.resize(n_targets);
  for (size_t i = 0; i < n_targets; i++) c.rep[i] = dsu.find((uint32_t)i);
  {
    // next wire in the same partition, in (row, column) scan order, cyclic
    std::vector<uint32_t> first(n_targets, UINT32_MAX), last(n_targets, UINT32_MAX);
    std::vector<uint32_t> next_wire(n * RW);  // routed-wire id = row*RW + col
    for (size_t row = 0; row < n; row++)
      for (int col = 0; col < RW; col++) {
        uint32_t wid = (uint32_t)(row * RW + col);
        uint32_t r = c.rep[row * W + col];
        if (first[r] == UINT32_MAX)
          first[r] = wid;
        else
          next_wire[last[r]] = wid;
        last[r] = wid;
      }
    for (size_t row = 0; row < n; row++)
      for (int col = 0; col < RW; col++) {
        uint32_t r = c.rep[row * W + col];
        uint32_t wid = (uint32_t)(row * RW + col);
        if (last[r] == wid) next_wire[wid] = first[r];
      }
    std::vector<u64> subgroup(n);
    u64 w = gl::root_of_unity(degree_bits), x = 1;
    for (size_t i = 0; i < n; i++) {
      subgroup[i] = x;
      x = gl::mul(x, w);
    }
    for (int col = 0; col < RW; col++) {
      std::vector<u64> s(n);
      for (size_t row = 0; row < n; row++) {
        uint32_t nb = next_wire[row * RW + col];
        size_t nrow = nb / RW, ncol = nb % RW;
        s[row] = gl::mul(c.k_is[ncol], subgroup[nrow]);
      }
      c.constants_sigmas.push_back(std::move(s));
    }
  }

  // generators: explicit ones first, then per-row gate generators (unused slots of incomplete
  // multi-op rows dropped, as upstream does with `incomplete_gates`)
  std::map<int, int> incomplete;  // row -> used ops
  for (auto& [key, rs] : current_slots_) incomplete[rs.first] = rs.second;
  c.generators = generators_;
  for (size_t row = 0; row < n; row++) {
    const GateInstance& gi = rows_[row];
    int ops = gate_generator_ops(gi.kind);
    auto it = incomplete.find((int)row);
    if (it != incomplete.end() && ops > 1) ops = it->second;
    for (int i = 0; i < ops; i++) c.generators.push_back(gate_op_generator(gi.kind, gi.constants, (int)row, i));
  }
  return c;
}

std::map<std::string, size_t> Circuit::gate_counts() const {
  std::map<std::string, size_t> m;
  for (auto& r : rows) m[gate_info(r.kind).id]++;
  return m;
}

}  // namespace p25
