// See p3_circuit.h.  Each function cites the reference lines it restates.  Statement order follows
// the Rust source exactly (Rust evaluates arguments left to right), because gate placement --
// and therefore the circuit -- depends on call order.
#include "p3_circuit.h"
#include <algorithm>
#include <stdexcept>

namespace p25 {

namespace {
constexpr int WIDTH = 12, DIGEST_ELEMS = 4, RATE = 4;  // src/p3/constants.rs
constexpr u64 TWO_ADIC_BASE = 1753635133440165772ULL;

int log2_strict(size_t n) {
  int r = 0;
  while (((size_t)1 << r) < n) r++;
  if (((size_t)1 << r) != n) throw std::logic_error("not a power of two");
  return r;
}
int log2_ceil(size_t n) {
  int r = 0;
  while (((size_t)1 << r) < n) r++;
  return r;
}
}  // namespace

// ---------------------------------------------------------------- src/p3/mod.rs:51-147
Target p3_constant(CircuitBuilder& cb, u64 v) { return cb.constant(v % gl::P); }

static Target combine_u64(CircuitBuilder& cb, std::array<Target, 2> lh) {
  return cb.mul_const_add((u64)1 << 32, lh[1], lh[0]);
}
Target p3_and(CircuitBuilder& cb, Target x, Target y) {
  auto [x_low, x_high] = cb.split_low_high(x, 32, 64);
  auto [y_low, y_high] = cb.split_low_high(y, 32, 64);
  auto r = cb.and_u64({x_low, x_high}, {y_low, y_high});
  return combine_u64(cb, r);
}
Target p3_xor(CircuitBuilder& cb, Target x, Target y) {
  auto [x_low, x_high] = cb.split_low_high(x, 32, 64);
  auto [y_low, y_high] = cb.split_low_high(y, 32, 64);
  auto r = cb.xor_u64({x_low, x_high}, {y_low, y_high});
  return combine_u64(cb, r);
}
Target p3_lsh(CircuitBuilder& cb, Target x, int n) {
  auto [x_low, x_high] = cb.split_low_high(x, 32, 64);
  auto r = cb.lsh_u64({x_low, x_high}, n);
  return combine_u64(cb, r);
}
Target p3_rsh(CircuitBuilder& cb, Target x, int n) {
  auto [x_low, x_high] = cb.split_low_high(x, 32, 64);
  auto r = cb.rsh_u64({x_low, x_high}, n);
  return combine_u64(cb, r);
}
// mod.rs:128-136 with binary_u32.rs:39-75 (convert_u32_bin32 = split_le 32, reverse_bin64,
// convert_bin32_u32 = le_sum)
Target reverse_p3(CircuitBuilder& cb, Target x) {
  auto [x_low, x_high] = cb.split_low_high(x, 32, 64);
  std::vector<BoolTarget> low_bits = cb.split_le(x_low, 32);
  std::vector<BoolTarget> high_bits = cb.split_le(x_high, 32);
  std::vector<BoolTarget> low_rev(high_bits.rbegin(), high_bits.rend());
  std::vector<BoolTarget> high_rev(low_bits.rbegin(), low_bits.rend());
  Target low_u32 = cb.le_sum(low_rev);
  Target high_u32 = cb.le_sum(high_rev);
  return cb.mul_const_add((u64)1 << 32, high_u32, low_u32);
}
Target reverse_p3_bits_len(CircuitBuilder& cb, Target x, int bit_len) {
  Target r = reverse_p3(cb, x);
  return p3_rsh(cb, r, 64 - bit_len);
}
static std::array<Target, 12> p3_arr12(CircuitBuilder& cb) {
  std::array<Target, 12> a;
  for (auto& t : a) t = cb.zero();
  return a;
}
static Ext p3_field_to_arr(CircuitBuilder& cb, Target x) {
  Ext r = {cb.zero(), cb.zero()};
  r[0] = x;
  return r;
}

// ---------------------------------------------------------------- src/p3/extension.rs
static Target p3_w(CircuitBuilder& cb) { return p3_constant(cb, 7); }
static Target p3_two_adic_generator(CircuitBuilder& cb, int bits) {
  Target base = p3_constant(cb, TWO_ADIC_BASE);
  return cb.exp_power_of_2(base, 32 - bits);
}
static Ext p3_ext_two_adic_generator(CircuitBuilder& cb, int bits) {
  Target base = p3_constant(cb, TWO_ADIC_BASE);
  Target x = cb.exp_power_of_2(base, 32 - bits);
  return p3_field_to_arr(cb, x);  // bits == 33 branch unreachable
}
static Ext p3_ext_one(CircuitBuilder& cb) {
  Target one = p3_constant(cb, 1);
  return p3_field_to_arr(cb, one);
}
static Ext p3_ext_zero(CircuitBuilder& cb) {
  Target zero = p3_constant(cb, 0);
  return p3_field_to_arr(cb, zero);
}
static Ext p3_ext_if(CircuitBuilder& cb, BoolTarget cond, Ext x, Ext y) {
  Ext res = {cb.zero(), cb.zero()};
  for (int i = 0; i < 2; i++) res[i] = cb._if(cond, x[i], y[i]);
  return res;
}
static Ext p3_ext_neg(CircuitBuilder& cb, Ext x) {
  for (auto& r : x) r = cb.neg(r);
  return x;
}
static Ext p3_ext_add(CircuitBuilder& cb, Ext x, Ext y) {
  for (int i = 0; i < 2; i++) x[i] = cb.add(x[i], y[i]);
  return x;
}
static Ext p3_ext_add_single(CircuitBuilder& cb, Ext x, Target y) {
  x[0] = cb.add(x[0], y);
  return x;
}
static Ext p3_ext_sub(CircuitBuilder& cb, Ext x, Ext y) {
  for (int i = 0; i < 2; i++) x[i] = cb.sub(x[i], y[i]);
  return x;
}
static Ext p3_ext_sub_single(CircuitBuilder& cb, Ext x, Target y) {
  x[0] = cb.sub(x[0], y);
  return x;
}
static Ext p3_ext_mul_single(CircuitBuilder& cb, const Ext& x, Target y) {
  Ext r;
  for (int i = 0; i < 2; i++) r[i] = cb.mul(x[i], y);
  return r;
}
// extension.rs:446-471 (EXT_DEGREE = 2 arm)
static Ext p3_ext_mul(CircuitBuilder& cb, const Ext& x, const Ext& y) {
  Target w_af = p3_w(cb);
  Ext res = {cb.zero(), cb.zero()};
  Target a0b0 = cb.mul(x[0], y[0]);
  Target w_b1 = cb.mul(w_af, y[1]);
  Target a1_w_b1 = cb.mul(x[1], w_b1);
  Target a0b1 = cb.mul(x[0], y[1]);
  Target a1b0 = cb.mul(x[1], y[0]);
  res[0] = cb.add(a0b0, a1_w_b1);
  res[1] = cb.add(a0b1, a1b0);
  return res;
}
// extension.rs:298-321 (EXT_DEGREE = 2 arm)
static Ext p3_ext_inverse(CircuitBuilder& cb, Ext a) {
  Target w = p3_w(cb);
  Target a0_sq = cb.square(a[0]);
  Target a1_sq = cb.square(a[1]);
  Target w_a1_sq = cb.mul(w, a1_sq);
  Target norm = cb.sub(a0_sq, w_a1_sq);
  Target scalar = cb.inverse(norm);
  Target a0s = cb.mul(a[0], scalar);
  Target a1_neg = cb.neg(a[1]);
  Target a1ns = cb.mul(a1_neg, scalar);
  Ext value = {cb.zero(), cb.zero()};
  value[0] = a0s;
  value[1] = a1ns;
  return value;
}
static Ext p3_ext_div(CircuitBuilder& cb, Ext x, Ext y) {
  Ext y_inv = p3_ext_inverse(cb, y);
  return p3_ext_mul(cb, y_inv, x);
}
static Ext p3_ext_exp_power_of_2(CircuitBuilder& cb, Ext x, int power_log) {
  Ext res = x;
  for (int i = 0; i < power_log; i++) res = p3_ext_mul(cb, res, res);
  return res;
}
static Ext p3_ext_monomial(CircuitBuilder& cb, int exponent) {
  Ext v = {cb.zero(), cb.zero()};
  v[exponent] = cb.one();
  return v;
}
static Ext p3_ext_mul_add(CircuitBuilder& cb, Ext x, Ext y, Ext z) {
  Ext xy = p3_ext_mul(cb, x, y);
  return p3_ext_add(cb, xy, z);
}
static void connect_p3_ext(CircuitBuilder& cb, const Ext& x, const Ext& y) {
  for (int i = 0; i < 2; i++) cb.connect(x[i], y[i]);
}

// ---------------------------------------------------------------- src/p3/challenger.rs
namespace {
struct DuplexChallengerTarget {
  std::vector<Target> sponge_state, input_buffer, output_buffer;
};
void p3_duplexing(CircuitBuilder& cb, DuplexChallengerTarget& x) {
  if (x.input_buffer.size() > (size_t)WIDTH) throw std::logic_error("challenger overflow");
  for (size_t i = 0; i < x.input_buffer.size(); i++) x.sponge_state[i] = x.input_buffer[i];
  x.input_buffer.clear();
  std::array<Target, 12> st;
  std::copy(x.sponge_state.begin(), x.sponge_state.end(), st.begin());
  auto out = cb.poseidon2_permute_targets(st);
  x.sponge_state.assign(out.begin(), out.end());
  x.output_buffer = x.sponge_state;
}
void p3_observe_single(CircuitBuilder& cb, DuplexChallengerTarget& x, Target v) {
  x.output_buffer.clear();
  x.input_buffer.push_back(v);
  if (x.input_buffer.size() == (size_t)WIDTH) p3_duplexing(cb, x);
}
template <class It>
void p3_observe(CircuitBuilder& cb, DuplexChallengerTarget& x, It b, It e) {
  for (; b != e; ++b) p3_observe_single(cb, x, *b);
}
Target p3_sample(CircuitBuilder& cb, DuplexChallengerTarget& x) {
  if (!x.input_buffer.empty() || x.output_buffer.empty()) p3_duplexing(cb, x);
  Target t = x.output_buffer.back();
  x.output_buffer.pop_back();
  return t;
}
Ext p3_sample_ext(CircuitBuilder& cb, DuplexChallengerTarget& x) {
  Target a = p3_sample(cb, x);
  Target b = p3_sample(cb, x);
  return Ext{a, b};
}
// challenger.rs:126-148
Target p3_sample_bits(CircuitBuilder& cb, DuplexChallengerTarget& x, int bits) {
  Target rand_f = p3_sample(cb, x);
  auto [rl, rh] = cb.split_low_high(rand_f, 32, 64);
  Target one = cb.one();
  Target power_of_bits = p3_constant(cb, (u64)1 << bits);
  Target pm1 = cb.sub(power_of_bits, one);
  auto [pl, ph] = cb.split_low_high(pm1, 32, 64);
  auto r = cb.and_u64({rl, rh}, {pl, ph});
  return cb.mul_const_add((u64)1 << 32, r[1], r[0]);
}
void p3_check_witness(CircuitBuilder& cb, DuplexChallengerTarget& x, int bits, Target witness) {
  p3_observe_single(cb, x, witness);
  Target res = p3_sample_bits(cb, x, bits);
  Target zero = cb.zero();
  cb.connect(res, zero);
}

// ---------------------------------------------------------------- src/p3/commit.rs
struct Dimensions {
  size_t width, height;
};
std::array<Target, 4> hash_iter_slices(CircuitBuilder& cb, const std::vector<const std::vector<Target>*>& slices) {
  auto state = p3_arr12(cb);
  std::vector<Target> flat;
  for (auto* s : slices) flat.insert(flat.end(), s->begin(), s->end());
  for (size_t off = 0; off < flat.size(); off += RATE) {
    size_t m = std::min((size_t)RATE, flat.size() - off);
    for (size_t i = 0; i < m; i++) state[i] = flat[off + i];
    state = cb.poseidon2_permute_targets(state);
  }
  return {state[0], state[1], state[2], state[3]};
}
std::array<Target, 4> compress(CircuitBuilder& cb, const std::array<Target, 4>& l, const std::array<Target, 4>& r) {
  auto state = p3_arr12(cb);
  for (int i = 0; i < 4; i++) {
    state[i] = l[i];
    state[4 + i] = r[i];
  }
  state = cb.poseidon2_permute_targets(state);
  return {state[0], state[1], state[2], state[3]};
}
size_t npo2(size_t x) {
  size_t r = 1;
  while (r < x) r <<= 1;
  return r;
}
// commit.rs:62-129
void verify_batch(CircuitBuilder& cb, const std::array<Target, 4>& commit, const std::vector<Dimensions>& dims,
                  Target index, const std::vector<std::vector<Target>>& opened_values,
                  const std::vector<std::array<Target, 4>>& proof) {
  std::vector<size_t> order(dims.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return dims[a].height > dims[b].height; });
  size_t pos = 0;
  size_t curr_height_padded = npo2(dims[order[0]].height);
  std::vector<const std::vector<Target>*> sl;
  while (pos < order.size() && npo2(dims[order[pos]].height) == curr_height_padded) sl.push_back(&opened_values[order[pos++]]);
  auto root = hash_iter_slices(cb, sl);
  for (const auto& sibling : proof) {
    Target one = cb.one();
    Target index_and_one = p3_and(cb, index, one);
    BoolTarget is_odd = index_and_one;
    std::array<Target, 4> left, right;
    for (int i = 0; i < DIGEST_ELEMS; i++) {
      left[i] = cb.zero();
      right[i] = cb.zero();
    }
    for (int i = 0; i < DIGEST_ELEMS; i++) {
      left[i] = cb._if(is_odd, sibling[i], root[i]);
      right[i] = cb._if(is_odd, root[i], sibling[i]);
    }
    root = compress(cb, left, right);
    index = p3_rsh(cb, index, 1);
    curr_height_padded >>= 1;
    if (pos < order.size() && npo2(dims[order[pos]].height) == curr_height_padded) {
      size_t next_height = dims[order[pos]].height;
      std::vector<const std::vector<Target>*> s2;
      while (pos < order.size() && dims[order[pos]].height == next_height) s2.push_back(&opened_values[order[pos++]]);
      auto d = hash_iter_slices(cb, s2);
      root = compress(cb, root, d);
    }
  }
  for (int i = 0; i < 4; i++) cb.connect(commit[i], root[i]);
}

// ---------------------------------------------------------------- src/p3/serde/two_adic.rs
struct Coset {
  int log_n;
  Target shift;
  size_t size() const { return (size_t)1 << log_n; }
};
Target coset_gen(CircuitBuilder& cb, const Coset& d) {
  Target base = cb.constant(TWO_ADIC_BASE);
  return cb.exp_power_of_2(base, 32 - d.log_n);
}
Ext coset_next_point(CircuitBuilder& cb, const Coset& d, Ext x) {
  Target g = coset_gen(cb, d);
  return p3_ext_mul_single(cb, x, g);
}
Ext coset_zp_at_point(CircuitBuilder& cb, const Coset& d, Ext point) {
  Target shift_inv = cb.inverse(d.shift);
  Ext p = p3_ext_mul_single(cb, point, shift_inv);
  Ext pw = p3_ext_exp_power_of_2(cb, p, d.log_n);
  Ext one = p3_ext_one(cb);
  return p3_ext_sub(cb, pw, one);
}
Target coset_zp_at_single_point(CircuitBuilder& cb, const Coset& d, Target point) {
  Target shift_inv = cb.inverse(d.shift);
  Target p = cb.mul(shift_inv, point);
  Target pw = cb.exp_power_of_2(p, d.log_n);
  Target one = cb.one();
  return cb.sub(pw, one);
}
struct LagrangeSelectors {
  Ext is_first_row, is_last_row, is_transition, inv_zeroifier;
};
// two_adic.rs:92-122
LagrangeSelectors selectors_at_point(CircuitBuilder& cb, const Coset& d, Ext point) {
  Target shift_inv = cb.inverse(d.shift);
  Ext unshifted = p3_ext_mul_single(cb, point, shift_inv);
  Ext un_pow = p3_ext_exp_power_of_2(cb, unshifted, d.log_n);
  Ext one = p3_ext_one(cb);
  Ext z_h = p3_ext_sub(cb, un_pow, one);
  Ext un_m1 = p3_ext_sub(cb, unshifted, one);
  Ext first = p3_ext_div(cb, z_h, un_m1);
  Target g = coset_gen(cb, d);
  Target g_inv = cb.inverse(g);
  Ext un_m_ginv = p3_ext_sub_single(cb, unshifted, g_inv);
  Ext last = p3_ext_div(cb, z_h, un_m_ginv);
  LagrangeSelectors s;
  s.is_first_row = first;
  s.is_last_row = last;
  s.is_transition = un_m_ginv;
  s.inv_zeroifier = p3_ext_inverse(cb, z_h);
  return s;
}
}  // namespace

// ---------------------------------------------------------------- src/p3/air.rs
void VerifierConstraintFolder::assert_zero(CircuitBuilder& cb, Ext x) {
  accumulator = p3_ext_mul_add(cb, accumulator, alpha, x);
}
void VerifierConstraintFolder::assert_eq(CircuitBuilder& cb, Ext x, Ext y) {
  Ext d = p3_ext_sub(cb, x, y);
  assert_zero(cb, d);
}
void VerifierConstraintFolder::when_assert_eq(CircuitBuilder& cb, Ext condition, Ext x, Ext y) {
  Ext d = p3_ext_sub(cb, x, y);
  Ext f = p3_ext_mul(cb, condition, d);
  assert_zero(cb, f);
}
// src/p3/mod.rs:176-221
void FibonacciAir::eval(VerifierConstraintFolder& folder, CircuitBuilder& cb) const {
  Ext la = folder.trace_local[0], lb = folder.trace_local[1], lc = folder.trace_local[2];
  Ext na = folder.trace_next[0], nb = folder.trace_next[1];
  Ext a_plus_b = p3_ext_add(cb, la, lb);
  folder.assert_eq(cb, a_plus_b, lc);
  Ext one = p3_ext_one(cb);
  folder.when_assert_eq(cb, folder.is_first_row, one, la);
  folder.when_assert_eq(cb, folder.is_first_row, one, lb);
  folder.when_assert_eq(cb, folder.is_transition, na, lb);
  folder.when_assert_eq(cb, folder.is_transition, nb, lc);
}

// ---------------------------------------------------------------- AIR programs
int AirProgram::node_degree(uint32_t i) const {
  std::vector<int> deg(i + 1, 0);
  for (uint32_t k = 0; k <= i; k++) {
    const Node& nd = nodes[k];
    if (nd.op == LOCAL || nd.op == NEXT) deg[k] = 1;
    else if (nd.op == CONST) deg[k] = 0;
    else if (nd.op == MUL) deg[k] = deg[nd.a] + deg[nd.b];
    else deg[k] = std::max(deg[nd.a], deg[nd.b]);
  }
  return deg[i];
}
void AirProgram::validate() const {
  if (width < 1 || width > 1024) throw std::invalid_argument("AIR: width out of range");
  if (nodes.empty() || constraints.empty()) throw std::invalid_argument("AIR: empty program");
  for (size_t i = 0; i < nodes.size(); i++) {
    const Node& nd = nodes[i];
    if (nd.op > MUL) throw std::invalid_argument("AIR: unknown node op");
    if ((nd.op == LOCAL || nd.op == NEXT) && nd.a >= (uint32_t)width) throw std::invalid_argument("AIR: column out of range");
    if (nd.op >= ADD && (nd.a >= i || nd.b >= i)) throw std::invalid_argument("AIR: node refers to a later node");
    if (nd.op == CONST && nd.value >= gl::P) throw std::invalid_argument("AIR: non-canonical constant");
  }
  for (const Constraint& c : constraints) {
    if (c.node >= nodes.size() || c.when > TRANSITION) throw std::invalid_argument("AIR: bad constraint");
    // degree <= 2: one quotient chunk, the reference's proof model (serde/proof.rs:41-48).  Degree 3: two chunks -- the
    // reference's verifier (verifier.rs:115-221) and its P3Config (mod.rs:76) already handle any power of two, only
    // `OpenedValues::add_virtual_to` fixes the count; round 5 lifted that.  Degree 4..5 (four chunks) needs log_blowup >= 2,
    // 6..9 (eight) log_blowup 3: the quotient domain 7*H_{n 2^lqd} must lie inside the LDE domain (round 6).
    if (node_degree(c.node) + (c.when == ALWAYS ? 0 : 1) > 9)
      throw std::invalid_argument("AIR: constraint degree > 9 needs more than eight quotient chunks");
  }
}
int AirProgram::max_constraint_degree() const {
  int d = 1;
  for (const Constraint& c : constraints) d = std::max(d, node_degree(c.node) + (c.when == ALWAYS ? 0 : 1));
  return d;
}
int AirProgram::log_quotient_degree() const {
  const int d = std::max(max_constraint_degree(), 2) - 1;
  int l = 0;
  while ((1 << l) < d) l++;
  return l;
}
AirProgram AirProgram::fibonacci() {
  AirProgram p;
  p.width = 3;
  auto nd = [&](uint32_t op, uint32_t a, uint32_t b, u64 v) {
    p.nodes.push_back(Node{op, a, b, v});
    return (uint32_t)p.nodes.size() - 1;
  };
  uint32_t la = nd(LOCAL, 0, 0, 0), lb = nd(LOCAL, 1, 0, 0), lc = nd(LOCAL, 2, 0, 0);
  uint32_t na = nd(NEXT, 0, 0, 0), nb = nd(NEXT, 1, 0, 0);
  uint32_t s = nd(ADD, la, lb, 0);
  uint32_t c0 = nd(SUB, s, lc, 0);
  uint32_t one = nd(CONST, 0, 0, 1);
  uint32_t c1 = nd(SUB, one, la, 0), c2 = nd(SUB, one, lb, 0);
  uint32_t c3 = nd(SUB, na, lb, 0), c4 = nd(SUB, nb, lc, 0);
  p.constraints = {{c0, ALWAYS}, {c1, FIRST_ROW}, {c2, FIRST_ROW}, {c3, TRANSITION}, {c4, TRANSITION}};
  return p;
}
namespace {
struct CircuitExtOps {
  CircuitBuilder& cb;
  Ext cst(u64 v) { return p3_field_to_arr(cb, p3_constant(cb, v)); }
  Ext add(const Ext& x, const Ext& y) { return p3_ext_add(cb, x, y); }
  Ext sub(const Ext& x, const Ext& y) { return p3_ext_sub(cb, x, y); }
  Ext mul(const Ext& x, const Ext& y) { return p3_ext_mul(cb, x, y); }
};
}  // namespace
void ProgramAir::eval(VerifierConstraintFolder& folder, CircuitBuilder& cb) const {
  CircuitExtOps ops{cb};
  const Ext sel[4] = {Ext{}, folder.is_first_row, folder.is_last_row, folder.is_transition};
  prog.fold<Ext>(folder.trace_local, folder.trace_next, sel, ops, [&](const Ext& c) { folder.assert_zero(cb, c); });
}

// ---------------------------------------------------------------- src/p3/serde/proof.rs:357-373
size_t P3Config::num_inputs() const {
  const size_t chunks = (size_t)1 << log_quotient_degree;   // 1 for the reference's proofs (proof.rs:41-48)
  size_t n = 8 + (size_t)trace_width * 4 + 4 * chunks;
  n += (size_t)log_trace_height * 4;
  size_t per_query = 0;
  // round i's layer has 2^(log_max_height - i) values committed in pairs: Merkle paths of log_max_height - 1 - i digests
  // (= log_trace_height - i for the reference's log_blowup 1: proof.rs:204-205)
  for (int i = 0; i < log_trace_height; i++) per_query += 2 + 4 * (size_t)(opening_matrix_log_max_height - 1 - i);
  n += per_query * fri_config.num_queries + 3;
  n += (size_t)fri_config.num_queries *
       ((trace_width + 4 * opening_matrix_log_max_height) +
        (chunks * opening_proof_query_openings_opened_values_length + 4 * opening_matrix_log_max_height));
  return n;
}
static P3ProofTarget add_virtual_proof(CircuitBuilder& cb, const P3Config& cfg) {
  std::vector<Target>& in = cb.input_targets;
  auto vt = [&]() {
    Target t = cb.add_virtual_target();
    in.push_back(t);
    return t;
  };
  auto v4 = [&]() { return std::array<Target, 4>{vt(), vt(), vt(), vt()}; };
  auto vext = [&]() { return Ext{vt(), vt()}; };
  P3ProofTarget p;
  p.trace_commit = v4();
  p.quotient_commit = v4();
  for (int i = 0; i < cfg.trace_width; i++) p.trace_local.push_back(vext());
  for (int i = 0; i < cfg.trace_width; i++) p.trace_next.push_back(vext());
  for (int ch = 0; ch < (1 << cfg.log_quotient_degree); ch++) {   // proof.rs:41-48 has `(0..1)`: one chunk; see AirProgram::validate
    Ext a = vext();
    Ext b = vext();
    p.quotient_chunks.push_back({a, b});
  }
  for (int i = 0; i < cfg.log_trace_height; i++) p.commit_phase_commits.push_back(v4());
  for (int q = 0; q < cfg.fri_config.num_queries; q++) {
    std::vector<P3CommitPhaseStep> steps;
    for (int i = 0; i < cfg.log_trace_height; i++) {
      P3CommitPhaseStep s;
      s.sibling_value = vext();
      for (int k = 0; k < cfg.opening_matrix_log_max_height - 1 - i; k++) s.opening_proof.push_back(v4());
      steps.push_back(std::move(s));
    }
    p.query_proofs.push_back(std::move(steps));
  }
  p.final_poly = vext();
  p.pow_witness = vt();
  for (int q = 0; q < cfg.fri_config.num_queries; q++) {
    std::array<P3BatchOpening, 2> bo;
    int widths[2] = {cfg.trace_width, cfg.opening_proof_query_openings_opened_values_length};
    for (int b = 0; b < 2; b++) {
      const int mats = b == 0 ? 1 : (1 << cfg.log_quotient_degree);   // batch 1: one matrix per quotient chunk
      for (int m = 0; m < mats; m++) {
        std::vector<Target> row;
        for (int i = 0; i < widths[b]; i++) row.push_back(vt());
        bo[b].opened_values.push_back(row);
      }
      for (int k = 0; k < cfg.opening_matrix_log_max_height; k++) bo[b].opening_proof.push_back(v4());
    }
    p.query_openings.push_back(std::move(bo));
  }
  p.degree_bits = cfg.degree_bits;
  return p;
}

// ---------------------------------------------------------------- src/p3/verifier.rs
namespace {
struct MatPoints {
  Coset domain;
  std::vector<std::pair<Ext, std::vector<Ext>>> points_and_values;
};
struct CommitAndPoints {
  std::array<Target, 4> commit;
  std::vector<MatPoints> mats;
};

// verifier.rs:424-519
Ext p3_verify_query(CircuitBuilder& cb, const std::vector<std::array<Target, 4>>& commit_phase_commits,
                    Target index, const std::vector<P3CommitPhaseStep>& steps, const std::vector<Ext>& betas,
                    const std::vector<Ext>& reduced_openings, int log_max_height) {
  Ext folded_eval = p3_ext_zero(cb);
  Target two_adic_generator = p3_two_adic_generator(cb, log_max_height);
  Target rev_index_shifted = reverse_p3_bits_len(cb, index, log_max_height);
  Target x0 = cb.exp(two_adic_generator, rev_index_shifted, 64);
  Ext x = p3_field_to_arr(cb, x0);
  Target one = cb.one();
  size_t n_steps = std::min({(size_t)log_max_height, commit_phase_commits.size(), steps.size(), betas.size()});
  for (size_t s = 0; s < n_steps; s++) {
    int log_folded_height = log_max_height - 1 - (int)s;
    const auto& commit = commit_phase_commits[s];
    const P3CommitPhaseStep& step = steps[s];
    const Ext& beta = betas[s];
    folded_eval = p3_ext_add(cb, reduced_openings[log_folded_height + 1], folded_eval);
    Target index_sibling = p3_xor(cb, index, one);
    Target index_pair = p3_rsh(cb, index, 1);
    Target isao = p3_and(cb, index_sibling, one);
    BoolTarget is_odd = isao;
    Ext evals[2] = {folded_eval, folded_eval};
    evals[0] = p3_ext_if(cb, is_odd, evals[0], step.sibling_value);
    evals[1] = p3_ext_if(cb, is_odd, step.sibling_value, evals[1]);
    std::vector<Dimensions> dims = {{2 * 2, (size_t)1 << log_folded_height}};
    std::vector<std::vector<Target>> ov = {{evals[0][0], evals[0][1], evals[1][0], evals[1][1]}};
    verify_batch(cb, commit, dims, index_pair, ov, step.opening_proof);

    Ext xs[2] = {x, x};
    Ext tag = p3_ext_two_adic_generator(cb, 1);
    Ext xs0g = p3_ext_mul(cb, xs[0], tag);
    Ext xs1g = p3_ext_mul(cb, xs[1], tag);
    Target one2 = cb.one();
    Target isao2 = p3_and(cb, index_sibling, one2);
    BoolTarget is_odd2 = isao2;
    xs[0] = p3_ext_if(cb, is_odd2, xs[0], xs0g);
    xs[1] = p3_ext_if(cb, is_odd2, xs1g, xs[1]);
    // interpolate and evaluate at beta
    Ext beta_minus_xs0 = p3_ext_sub(cb, beta, xs[0]);
    Ext e1_minus_e0 = p3_ext_sub(cb, evals[1], evals[0]);
    Ext xs1_minus_xs0 = p3_ext_sub(cb, xs[1], xs[0]);
    Ext num = p3_ext_mul(cb, e1_minus_e0, beta_minus_xs0);
    Ext q = p3_ext_div(cb, num, xs1_minus_xs0);
    folded_eval = p3_ext_add(cb, evals[0], q);
    index = index_pair;
    x = p3_ext_mul(cb, x, x);
  }
  return folded_eval;
}

// verifier.rs:242-355, 357-388, 390-422
void p3_verify_opening_proof(CircuitBuilder& cb, const P3FriConfig& config,
                             const std::vector<CommitAndPoints>& commits_and_points, const P3ProofTarget& proof,
                             DuplexChallengerTarget& challenger) {
  Ext alpha = p3_sample_ext(cb, challenger);
  // p3_verify_shape_and_sample_challenges
  std::vector<Ext> betas;
  for (const auto& comm : proof.commit_phase_commits) {
    p3_observe(cb, challenger, comm.begin(), comm.end());
    betas.push_back(p3_sample_ext(cb, challenger));
  }
  if ((int)proof.query_proofs.size() != config.num_queries) throw std::logic_error("InvalidProofShape");
  p3_check_witness(cb, challenger, config.proof_of_work_bits, proof.pow_witness);
  const int log_max_height = (int)proof.commit_phase_commits.size() + config.log_blowup;
  std::vector<Target> query_indices;
  for (int i = 0; i < config.num_queries; i++) query_indices.push_back(p3_sample_bits(cb, challenger, log_max_height));

  std::vector<std::vector<Ext>> reduced_openings;
  for (size_t qi = 0; qi < proof.query_openings.size() && qi < query_indices.size(); qi++) {
    const auto& query_opening = proof.query_openings[qi];
    Target index = query_indices[qi];
    std::vector<Ext> ro(32);
    for (auto& e : ro) e = Ext{cb.zero(), cb.zero()};
    Ext one = p3_ext_one(cb);
    std::vector<Ext> alpha_pow(32, one);
    for (size_t b = 0; b < 2 && b < commits_and_points.size(); b++) {
      const P3BatchOpening& batch_opening = query_opening[b];
      const CommitAndPoints& cp = commits_and_points[b];
      std::vector<Dimensions> batch_dims;
      for (const auto& m : cp.mats) batch_dims.push_back({0, m.domain.size()});
      verify_batch(cb, cp.commit, batch_dims, index, batch_opening.opened_values, batch_opening.opening_proof);
      for (size_t mi = 0; mi < batch_opening.opened_values.size() && mi < cp.mats.size(); mi++) {
        const std::vector<Target>& mat_opening = batch_opening.opened_values[mi];
        const MatPoints& mp = cp.mats[mi];
        int log_height = log2_strict(mp.domain.size()) + config.log_blowup;
        int bits_reduced = log_max_height - log_height;
        Target index_rs = p3_rsh(cb, index, bits_reduced);
        Target rev_reduced_index = reverse_p3_bits_len(cb, index_rs, log_height);
        Target generator = p3_w(cb);
        Target tag = p3_two_adic_generator(cb, log_height);
        Target tag_pow = cb.exp(tag, rev_reduced_index, 64);
        Target x = cb.mul(generator, tag_pow);
        for (const auto& [z, ps_at_z] : mp.points_and_values) {
          size_t cnt = std::min(mat_opening.size(), ps_at_z.size());
          for (size_t k = 0; k < cnt; k++) {
            Target p_at_x = mat_opening[k];
            const Ext& p_at_z = ps_at_z[k];
            Ext p_at_z_neg = p3_ext_neg(cb, p_at_z);
            Ext z_neg = p3_ext_neg(cb, z);
            Ext numer = p3_ext_add_single(cb, p_at_z_neg, p_at_x);
            Ext denom = p3_ext_add_single(cb, z_neg, x);
            Ext quotient = p3_ext_div(cb, numer, denom);
            Ext t = p3_ext_mul(cb, alpha_pow[log_height], quotient);
            ro[log_height] = p3_ext_add(cb, ro[log_height], t);
            alpha_pow[log_height] = p3_ext_mul(cb, alpha_pow[log_height], alpha);
          }
        }
      }
    }
    reduced_openings.push_back(std::move(ro));
  }
  // p3_verify_challenges
  for (size_t qi = 0; qi < query_indices.size() && qi < proof.query_proofs.size() && qi < reduced_openings.size(); qi++) {
    Ext folded = p3_verify_query(cb, proof.commit_phase_commits, query_indices[qi], proof.query_proofs[qi], betas,
                                 reduced_openings[qi], log_max_height);
    connect_p3_ext(cb, folded, proof.final_poly);
  }
}
}  // namespace

// verifier.rs:100-240 (__p3_verify_proof__) behind mod.rs:66-94 (p3_verify_proof)
P3ProofTarget p3_verify_proof(CircuitBuilder& cb, const P3Config& config, const Air& air) {
  DuplexChallengerTarget challenger;
  {
    auto st = p3_arr12(cb);
    challenger.sponge_state.assign(st.begin(), st.end());
  }
  P3ProofTarget proof = add_virtual_proof(cb, config);

  const int degree_bits = proof.degree_bits;
  const size_t degree = (size_t)1 << degree_bits;
  const size_t quotient_degree = (size_t)1 << config.log_quotient_degree;

  // TwoAdicMultiplicativeCoset::natural_domain_for_degree
  if (log2_strict(degree) > config.log_trace_height) throw std::logic_error("degree above trace height");
  Coset trace_domain{log2_strict(degree), cb.one()};
  // create_disjoint_domain
  Coset quotient_domain;
  {
    Target generator = cb.constant(7);
    quotient_domain.log_n = log2_ceil((size_t)1 << (degree_bits + config.log_quotient_degree));
    quotient_domain.shift = cb.mul(trace_domain.shift, generator);
  }
  // split_domains
  std::vector<Coset> quotient_chunks_domains;
  {
    int log_chunks = log2_strict(quotient_degree);
    Target g = coset_gen(cb, quotient_domain);
    for (size_t i = 0; i < quotient_degree; i++) {
      Target gi = cb.exp_u64(g, i);
      Target shift = cb.mul(quotient_domain.shift, gi);
      quotient_chunks_domains.push_back(Coset{quotient_domain.log_n - log_chunks, shift});
    }
  }
  const int air_width = air.width();
  bool valid_shape = (int)proof.trace_local.size() == air_width && (int)proof.trace_next.size() == air_width &&
                     proof.quotient_chunks.size() == quotient_degree;
  for (auto& qc : proof.quotient_chunks) valid_shape = valid_shape && qc.size() == 2;
  if (!valid_shape) throw std::logic_error("Invalid Proof Shape");  // verifier.rs:131-133

  p3_observe(cb, challenger, proof.trace_commit.begin(), proof.trace_commit.end());
  Ext alpha = p3_sample_ext(cb, challenger);
  p3_observe(cb, challenger, proof.quotient_commit.begin(), proof.quotient_commit.end());
  Ext zeta = p3_sample_ext(cb, challenger);
  Ext zeta_next = coset_next_point(cb, trace_domain, zeta);

  std::vector<CommitAndPoints> cps(2);
  cps[0].commit = proof.trace_commit;
  cps[0].mats.push_back(MatPoints{trace_domain, {{zeta, proof.trace_local}, {zeta_next, proof.trace_next}}});
  cps[1].commit = proof.quotient_commit;
  for (size_t i = 0; i < quotient_chunks_domains.size() && i < proof.quotient_chunks.size(); i++)
    cps[1].mats.push_back(MatPoints{quotient_chunks_domains[i], {{zeta, proof.quotient_chunks[i]}}});
  p3_verify_opening_proof(cb, config.fri_config, cps, proof, challenger);

  // zps (verifier.rs:169-198)
  std::vector<Ext> zps;
  for (size_t i = 0; i < quotient_chunks_domains.size(); i++) {
    const Coset& domain = quotient_chunks_domains[i];
    std::vector<Ext> terms;
    for (size_t j = 0; j < quotient_chunks_domains.size(); j++) {
      if (j == i) continue;
      const Coset& other = quotient_chunks_domains[j];
      Ext other_zeta = coset_zp_at_point(cb, other, zeta);
      Target first_point = domain.shift;
      Target other_first = coset_zp_at_single_point(cb, other, first_point);
      Target other_first_inv = cb.inverse(other_first);
      terms.push_back(p3_ext_mul_single(cb, other_zeta, other_first_inv));
    }
    // `.unwrap_or({..})` evaluates its argument eagerly
    Target one = cb.one();
    Ext dflt = p3_field_to_arr(cb, one);
    if (terms.empty()) {
      zps.push_back(dflt);
    } else {
      Ext acc = terms[0];
      for (size_t k = 1; k < terms.size(); k++) acc = p3_ext_mul(cb, acc, terms[k]);
      zps.push_back(acc);
    }
  }
  // quotient recomposition (verifier.rs:200-221)
  std::vector<Ext> per_chunk;
  for (size_t ch_i = 0; ch_i < proof.quotient_chunks.size(); ch_i++) {
    std::vector<Ext> parts;
    for (size_t e_i = 0; e_i < proof.quotient_chunks[ch_i].size(); e_i++) {
      Ext monomial = p3_ext_monomial(cb, (int)e_i);
      Ext mc = p3_ext_mul(cb, monomial, proof.quotient_chunks[ch_i][e_i]);
      parts.push_back(p3_ext_mul(cb, zps[ch_i], mc));
    }
    Ext acc = parts[0];
    for (size_t k = 1; k < parts.size(); k++) acc = p3_ext_add(cb, acc, parts[k]);
    per_chunk.push_back(acc);
  }
  Ext quotient = per_chunk[0];
  for (size_t k = 1; k < per_chunk.size(); k++) quotient = p3_ext_add(cb, quotient, per_chunk[k]);

  LagrangeSelectors sels = selectors_at_point(cb, trace_domain, zeta);
  VerifierConstraintFolder folder;
  folder.trace_local = proof.trace_local;
  folder.trace_next = proof.trace_next;
  folder.is_first_row = sels.is_first_row;
  folder.is_last_row = sels.is_last_row;
  folder.is_transition = sels.is_transition;
  folder.alpha = alpha;
  folder.accumulator = p3_ext_zero(cb);
  air.eval(folder, cb);
  Ext folded_constraints = folder.accumulator;
  Ext lhs = p3_ext_mul(cb, folded_constraints, sels.inv_zeroifier);
  connect_p3_ext(cb, lhs, quotient);
  return proof;
}


Circuit build_gadget_circuit(int kind, int param) {
  CircuitBuilder cb;
  auto in = [&]() {
    Target t = cb.add_virtual_target();
    cb.input_targets.push_back(t);
    return t;
  };
  switch (kind) {
    case GADGET_AND:
    case GADGET_XOR: {
      Target x = in(), y = in(), expected = in();
      Target r = kind == GADGET_AND ? p3_and(cb, x, y) : p3_xor(cb, x, y);
      cb.connect(r, expected);
      break;
    }
    case GADGET_LSH:
    case GADGET_RSH: {
      if (param < 0 || param > 63 || param == 32) throw std::invalid_argument("shift amount");
      Target x = in(), expected = in();
      Target r = kind == GADGET_LSH ? p3_lsh(cb, x, param) : p3_rsh(cb, x, param);
      cb.connect(r, expected);
      break;
    }
    case GADGET_REVERSE: {
      if (param < 1 || param > 64) throw std::invalid_argument("bit length");
      Target x = in(), expected = in();
      Target r = reverse_p3_bits_len(cb, x, param);
      cb.connect(r, expected);
      break;
    }
    case GADGET_COMPRESS: {  // MerkleTreeMmcs::compress (commit.rs:48-60) on 8 inputs
      std::array<Target, 4> l, rr;
      for (auto& t : l) t = in();
      for (auto& t : rr) t = in();
      auto out = compress(cb, l, rr);
      for (int i = 0; i < 4; i++) cb.connect(out[i], in());
      break;
    }
    case GADGET_EXP: {  // x = 7 * w^e as in verifier.rs:299-311
      Target e = in(), expected = in();
      Target g = p3_two_adic_generator(cb, param);
      Target pw = cb.exp(g, e, 64);
      Target x = cb.mul(p3_w(cb), pw);
      Target xi = cb.inverse(x);
      Target one = cb.mul(x, xi);
      cb.connect(one, cb.one());
      cb.connect(x, expected);
      break;
    }
    case GADGET_HASH_SLICES: {  // MerkleTreeMmcs::hash_iter_slices (commit.rs:23-46; test_hash_iter_slice :143-171)
      if (param < 1 || param > 16) throw std::invalid_argument("hash_iter_slices gadget: 1..16 slices of 4 words");
      std::vector<std::vector<Target>> slices(param, std::vector<Target>(4));
      for (auto& sl : slices)
        for (auto& t : sl) t = in();
      std::vector<const std::vector<Target>*> ptrs;
      for (auto& sl : slices) ptrs.push_back(&sl);
      auto out = hash_iter_slices(cb, ptrs);
      for (int i = 0; i < 4; i++) cb.connect(out[i], in());
      break;
    }
    case GADGET_CONNECTED_INPUTS: {  // two witness-set targets under one copy constraint, then a * b == expected
      Target a = in(), b = in(), expected = in();
      cb.connect(a, b);
      cb.connect(cb.mul(a, b), expected);
      break;
    }
    case GADGET_EXT_ARITH: {  // extension-field gadgets of upstream gadgets/arithmetic_extension.rs (recursion, 8f-4)
      Ext a = {in(), in()}, b = {in(), in()}, c = {in(), in()};
      Ext t = cb.mul_add_extension(a, b, c);                    // a*b + c          (ArithmeticExtensionGate)
      t = cb.sub_extension(cb.mul_extension(t, a), b);          // t*a - b          (MulExtensionGate, ArithmeticExtensionGate)
      t = cb.mul_const_add_extension(5, t, cb.constant_extension(gl::E2{3, 9}));
      Ext t7 = cb.exp_u64_extension(t, 7);
      Ext q = cb.div_extension(t7, c);                          // virtual inverse + QuotientGeneratorExtension
      Ext s = cb.add_many_extension({q, a, b, cb.scalar_mul_ext(a[0], b)});
      cb.connect_extension(s, Ext{in(), in()});
      break;
    }
    case GADGET_POSEIDON_MERKLE: {  // hash_or_noop of `param` words, then one Merkle step with a swap bit (PoseidonGate)
      if (param < 1 || param > 200) throw std::invalid_argument("poseidon gadget: 1..200 leaf words");
      std::vector<Target> leaf;
      for (int i = 0; i < param; i++) leaf.push_back(in());
      std::array<Target, 4> h = cb.hash_or_noop(leaf);
      std::array<Target, 4> sib;
      for (auto& t : sib) t = in();
      Target bit = in();
      cb.connect(cb.mul_sub(bit, bit, bit), cb.zero());        // bit is boolean
      std::array<Target, 12> st;
      for (int i = 0; i < 4; i++) {
        st[i] = h[i];
        st[4 + i] = sib[i];
        st[8 + i] = cb.zero();
      }
      auto out = cb.poseidon_permute_swapped(st, bit);
      for (int i = 0; i < 4; i++) cb.connect(out[i], in());
      break;
    }
    case GADGET_PUBLIC_INPUTS: {  // upstream `register_public_input(s)`: `param` inputs x_i; exposes every x_i, then their
      // running products x_0 x_1, x_0 x_1 x_2, ... (so public inputs are a mix of virtual targets and gate wires)
      if (param < 1 || param > 64) throw std::invalid_argument("public-inputs gadget: 1..64 inputs");
      std::vector<Target> xs;
      for (int i = 0; i < param; i++) xs.push_back(in());
      cb.register_public_inputs(xs);
      Target acc = xs[0];
      for (int i = 1; i < param; i++) {
        acc = cb.mul(acc, xs[i]);
        cb.register_public_input(acc);
      }
      break;
    }
    case GADGET_INTERLEAVE_U32: {  // the reference's test_interleave_u32 (gadgets/interleaved_u32.rs:354-382)
      // param 1: the test's circuit as written -- x = constant_u32(0xFFFFFFFC), no witness inputs; param 0: x is an input
      if (param != 0 && param != 1) throw std::invalid_argument("interleave gadget: param 0 (input) or 1 (the reference's constant)");
      Target x = param ? cb.constant_u32(0xFFFFFFFCu) : in();
      cb.register_public_input(cb.interleave_u32(x));
      break;
    }
    case GADGET_UNINTERLEAVE_TO_U32: {  // test_uninterleave_to_u32 (gadgets/interleaved_u32.rs:388-417)
      if (param != 0 && param != 1) throw std::invalid_argument("uninterleave gadget: param 0 (input) or 1 (the reference's constant)");
      Target x = param ? cb.constant(0xF555555555555555ull) : in();
      auto [evens, odds] = cb.uninterleave_to_u32(x);
      cb.register_public_input(evens);
      cb.register_public_input(odds);
      break;
    }
    case GADGET_REFERENCE_GATES: {  // all four gates the reference defines on this path + ArithmeticGate, in one small circuit:
      // the circuit tests/blob_writer.py builds INDEPENDENTLY from the blob specification (two selector groups, 8 gate types)
      Target x = in(), y = in(), z = in();
      Target m = cb.mul(x, y);                           // ArithmeticGate
      Target xi = cb.interleave_u32(x);                  // U32InterleaveGate, ops 0 and 1 of one row
      Target yi = cb.interleave_u32(y);
      auto [ev, od] = cb.uninterleave_to_u32(xi);        // UninterleaveToU32Gate
      auto [lo, hi] = cb.mul_add_u32(x, y, z);           // U32ArithmeticGate
      std::array<Target, 12> st = {m, xi, yi, ev, od, lo, hi, x, cb.zero(), cb.zero(), cb.zero(), cb.zero()};
      auto out = cb.poseidon2_permute_targets(st);       // Poseidon2Gate
      for (Target t : {m, xi, yi, ev, od, lo, hi}) cb.connect(t, in());
      for (int i = 0; i < 4; i++) cb.connect(out[i], in());
      break;
    }
    default:
      throw std::invalid_argument("unknown gadget kind");
  }
  return cb.build();
}

}  // namespace p25
