// Host-side circuit builder: the subset of upstream plonky2's `CircuitBuilder` (crate plonky2 @
// 3de92d9, absent from /root/reference; restated from SURVEY.md App. A and the upstream design)
// that the plonky3-verifier circuit of /root/reference/src/p3 touches, plus the reference's own
// gadget layers (src/common/u32/gadgets, src/common/poseidon2/poseidon2.rs:585-609, src/p3/*).
//
// It runs once per circuit SHAPE (not on the per-proof path) and produces the `Circuit` tables
// the GPU prover consumes: gate rows, constants, selectors, sigmas, the levelised witness-generator
// program and the input map.  Method names mirror the reference/upstream API so that the circuit
// emission code (p3_circuit.cpp) reads like src/p3/*.rs.
#pragma once
#include <stdint.h>
#include <array>
#include <map>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>
#include "gl.h"

namespace p25 {

struct Target {
  int32_t row;  // -1: virtual target
  int32_t col;  // wire column, or virtual index
  bool operator==(const Target& o) const { return row == o.row && col == o.col; }
  bool operator!=(const Target& o) const { return !(*this == o); }
  bool operator<(const Target& o) const { return row != o.row ? row < o.row : col < o.col; }
  bool is_virtual() const { return row < 0; }
};
inline Target wire(int row, int col) { return Target{row, col}; }
struct TargetHash {
  size_t operator()(const Target& t) const { return ((uint64_t)(uint32_t)t.row << 32 | (uint32_t)t.col) * 0x9E3779B97F4A7C15ull >> 7; }
};
typedef Target BoolTarget;
typedef std::array<Target, 2> Ext;  // p3 BinomialExtensionField<Target> / plonky2 ExtensionTarget<2>

// CircuitConfig::standard_recursion_config() (used at src/p3/mod.rs:231)
struct CircuitConfig {
  int num_wires = 135;
  int num_routed_wires = 80;
  int num_constants = 2;
  int num_challenges = 2;
  int max_quotient_degree_factor = 8;
  int rate_bits = 3;
  int cap_height = 4;
  int proof_of_work_bits = 16;
  int num_query_rounds = 28;
  int fri_arity_bits = 4;      // ConstantArityBits(4, 5)
  int fri_final_poly_bits = 5;
};

enum GateKind : uint8_t {
  G_NOOP = 0,
  G_CONSTANT,
  G_PUBLIC_INPUT,
  G_BASE_SUM,           // BaseSumGate<2>{num_limbs: 63}
  G_U32_INTERLEAVE,     // src/common/u32/gates/interleave_u32.rs
  G_U32_UNINTERLEAVE,   // src/common/u32/gates/uninterleave_to_u32.rs
  G_ARITHMETIC,         // ArithmeticGate{num_ops: 20}
  G_MUL_EXT,            // MulExtensionGate{num_ops: 13}
  G_EXPONENTIATION,     // ExponentiationGate{num_power_bits: 66}
  G_U32_ARITHMETIC,     // src/common/u32/gates/arithmetic_u32.rs
  G_POSEIDON2,          // src/common/poseidon2/poseidon2_gate.rs
  // recursion: the gates of upstream's recursive verifier used by recursion.cpp
  G_ARITH_EXT,          // ArithmeticExtensionGate{num_ops: 10}
  G_POSEIDON,           // PoseidonGate (Poseidon v1 permutation with swap; hashes and Merkle paths of inner proofs)
  // round 3: more of upstream's recursive-verifier gate set (also importable from upstream-built circuits)
  G_RANDOM_ACCESS,      // RandomAccessGate{bits: 4, num_copies: 4, num_extra_constants: 2}: list[index] of 16 elements
  G_REDUCING,           // ReducingGate{num_coeffs: 43}: acc_{i+1} = acc_i * alpha + coeff_i, base-field coefficients
  G_REDUCING_EXT,       // ReducingExtensionGate{num_coeffs: 32}: the same with extension-field coefficients
  G_COSET_INTERP,       // CosetInterpolationGate{subgroup_bits: 4, degree: 6}: interpolant of 16 values on shift*H at a point
  G_POSEIDON_MDS,       // PoseidonMdsGate: Poseidon's MDS layer on 12 extension elements (PoseidonGate evaluated in-circuit)
  G_NUM_KINDS
};
struct GateInfo {
  const char* id;  // upstream `Gate::id()` string: orders the gates (sorted by (degree, id))
  int degree, num_constants, num_constraints, num_ops;
};
const GateInfo& gate_info(GateKind k);
// limits of the device kernels, enforced when a circuit enters the library (circuit_io.cpp)
constexpr int ALPHA_POWS = 192;   // alpha-power table of the quotient kernel: max constraints per gate
constexpr int MAX_ROUTED = 128;   // routed wires the permutation argument kernels hold
constexpr int MAX_CHUNKS = 16;    // partial-product chunks per challenge (num_partial_products + 1) they hold per row
constexpr int MAX_PUBLIC_INPUTS = 4096;  // public inputs a circuit may register (hashed by one wave per proof)
constexpr int BASE_SUM_LIMBS = 63;
constexpr int EXP_POWER_BITS = 66;
// the instances of upstream's parameterised gates that `new_from_config` / `max_coeffs_len` give for the standard
// recursion config (135 wires, 80 routed, 2 constants): RandomAccessGate::new_from_config(config, 4),
// ReducingGate::max_coeffs_len = min(80 - 6, (135 - 4) / 3), ReducingExtensionGate: min((80 - 6) / 2, (135 - 4) / 4)
constexpr int RA_BITS = 4, RA_VEC = 16, RA_COPIES = 4, RA_EXTRA_CONSTS = 2, RA_ROUTED = (2 + RA_VEC) * RA_COPIES + RA_EXTRA_CONSTS;
constexpr int RED_COEFFS = 43, REDX_COEFFS = 32;
// CosetInterpolationGate::with_max_degree(4, max_quotient_degree_factor = 8): n_intermediates = (16 - 2) / (8 - 1) = 2,
// degree = (16 - 2) / (2 + 1) + 2 = 6.  Wires: shift 0 | values 1..32 | point 33,34 | value 35,36 | intermediate
// evals 37..40 | intermediate products 41..44 | shifted point 45,46.  Chunks of the barycentric recurrence: points
// [0,6), [6,11), [11,16).
constexpr int CI_POINTS = 16, CI_DEGREE = 6, CI_INTER = 2, CI_W_POINT = 33, CI_W_VALUE = 35, CI_W_INTER = 37,
              CI_W_SHIFTED = CI_W_INTER + 4 * CI_INTER, CI_WIRES = CI_W_SHIFTED + 2;
inline int ci_chunk_begin(int c) { return c == 0 ? 0 : 1 + (CI_DEGREE - 1) * c; }
inline int ci_chunk_end(int c) { return c == 0 ? CI_DEGREE : (1 + (CI_DEGREE - 1) * (c + 1) < CI_POINTS ? 1 + (CI_DEGREE - 1) * (c + 1) : CI_POINTS); }

enum GenKind : uint32_t {
  GEN_CONSTANT = 0,     // out = c0
  GEN_RANDOM,           // out = per-proof pseudo-random filler (upstream RandomValueGenerator)
  GEN_ARITHMETIC,       // ArithmeticBaseGenerator: out = c0*m0*m1 + c1*addend
  GEN_MUL_EXT,          // MulExtensionGenerator: out = c0 * m0 * m1 in F_p^2
  GEN_QUOTIENT_EXT,     // QuotientGeneratorExtension: out = num / den in F_p^2
  GEN_BASE_SPLIT,       // BaseSplitGenerator<2>: sum -> 63 limbs
  GEN_WIRE_SPLIT,       // WireSplitGenerator: integer -> per-gate 63-bit sums
  GEN_BASE_SUM,         // BaseSumGenerator<2>: limbs -> sum
  GEN_LOW_HIGH,         // LowHighGenerator: x -> (x mod 2^n_log, x >> n_log)
  GEN_EXPONENTIATION,   // ExponentiationGenerator
  GEN_POSEIDON2,        // Poseidon2Generator (poseidon2_gate.rs:447-523)
  GEN_U32_ARITHMETIC,   // arithmetic_u32.rs:389-439
  GEN_U32_INTERLEAVE,   // interleave_u32.rs:305-334
  GEN_U32_UNINTERLEAVE, // uninterleave_to_u32.rs:353-390
  GEN_ARITH_EXT,        // ArithmeticExtensionGenerator: out = c0*m0*m1 + c1*addend in F_p^2
  GEN_POSEIDON,         // PoseidonGenerator: deltas, S-box inputs, outputs of one Poseidon permutation
  GEN_RANDOM_ACCESS,    // RandomAccessGenerator: index, 16 items -> claimed element, 4 index bits
  GEN_REDUCING,         // ReducingGenerator: alpha, old_acc (ext), 43 base coefficients -> 43 accumulators (ext)
  GEN_REDUCING_EXT,     // ReducingGenerator of the extension gate: 32 ext coefficients -> 32 accumulators
  GEN_COSET_INTERP,     // InterpolationGenerator: shift, 16 ext values, point -> shifted point, 2 x (eval, prod), value
  GEN_POSEIDON_MDS,     // PoseidonMdsGenerator: 12 ext inputs -> 12 ext outputs
  GEN_NUM_KINDS
};
struct Generator {
  GenKind kind;
  u64 c0 = 0, c1 = 0;
  int aux = 0;  // n_log (LOW_HIGH)
  std::vector<Target> deps, outs;
};

struct GateInstance {
  GateKind kind;
  u64 constants[2];
};

struct Generator;
// witness generators of a gate row (upstream `Gate::generators`): count per row and the i-th one
int gate_generator_ops(GateKind k);
// Everything the prover needs about one circuit shape.
struct Circuit {
  CircuitConfig cfg;
  int degree_bits = 0;
  size_t degree() const { return (size_t)1 << degree_bits; }
  std::vector<GateInstance> rows;
  // gate types present, sorted by (degree, id); selector grouping (upstream selectors.rs)
  std::vector<GateKind> gates;
  std::vector<int> selector_index;              // per sorted gate
  std::vector<std::pair<int, int>> groups;      // [start, end) in sorted-gate indices
  int num_selectors = 0;
  int num_gate_constraints = 0;
  int num_partial_products = 0;
  std::vector<int> fri_reduction_arity_bits;
  // constants_sigmas polynomials in oracle order: selectors | constants | sigmas, each `degree` values
  std::vector<std::vector<u64>> constants_sigmas;
  std::vector<u64> k_is;
  // witness program
  size_t num_virtual_targets = 0;
  std::vector<Target> input_targets;            // per-proof inputs, in `add_virtual_to` order
  std::vector<Generator> generators;
  std::vector<uint32_t> rep;                    // representative map over target indices
  size_t target_index(Target t) const {
    return t.is_virtual() ? degree() * cfg.num_wires + t.col : (size_t)t.row * cfg.num_wires + t.col;
  }
  size_t num_targets() const { return degree() * cfg.num_wires + num_virtual_targets; }
  int pi_row = -1;
  // upstream `builder.register_public_input`: targets whose values the proof exposes (ProofWithPublicInputs::public_inputs);
  // their Poseidon hash is what the PublicInputGate row holds and what the transcript absorbs after the circuit digest
  std::vector<Target> public_inputs;
  // statistics
  std::map<std::string, size_t> gate_counts() const;
};

class CircuitBuilder {
 public:
  explicit CircuitBuilder(const CircuitConfig& cfg = CircuitConfig()) : config(cfg) {}
  CircuitConfig config;

  // ---- core (upstream circuit_builder.rs) ----
  Target add_virtual_target();
  std::vector<Target> add_virtual_targets(int n);
  Target constant(u64 c);
  Target zero() { return constant(0); }
  Target one() { return constant(1); }
  Target two() { return constant(2); }
  Target neg_one() { return constant(gl::P - 1); }
  BoolTarget _false() { return zero(); }
  BoolTarget _true() { return one(); }
  BoolTarget constant_bool(bool b) { return b ? _true() : _false(); }
  std::optional<u64> target_as_constant(Target t) const;
  void connect(Target x, Target y);
  void assert_zero(Target x) { connect(x, zero()); }
  int add_gate(GateKind k, u64 c0 = 0, u64 c1 = 0);
  std::pair<int, int> find_slot(GateKind k, int n_params, u64 p0, u64 p1);
  void add_generator(Generator g) { generators_.push_back(std::move(g)); }
  // upstream circuit_builder.rs `register_public_input(s)`
  void register_public_input(Target t) { public_inputs_.push_back(t); }
  void register_public_inputs(const std::vector<Target>& ts) { public_inputs_.insert(public_inputs_.end(), ts.begin(), ts.end()); }
  size_t num_gates() const { return rows_.size(); }

  // ---- base arithmetic (upstream gadgets/arithmetic.rs) ----
  Target arithmetic(u64 c0, u64 c1, Target m0, Target m1, Target addend);
  Target mul_add(Target x, Target y, Target z) { return arithmetic(1, 1, x, y, z); }
  Target mul_sub(Target x, Target y, Target z) { return arithmetic(1, gl::P - 1, x, y, z); }
  Target add(Target x, Target y) { return arithmetic(1, 1, x, one(), y); }
  Target sub(Target x, Target y) { return arithmetic(1, gl::P - 1, x, one(), y); }
  Target mul(Target x, Target y) { return arithmetic(1, 0, x, y, x); }
  Target neg(Target x) { return mul(x, neg_one()); }
  Target square(Target x) { return mul(x, x); }
  Target mul_const_add(u64 c, Target x, Target y) { Target ct = constant(c); return mul_add(ct, x, y); }
  Target exp_power_of_2(Target base, int power_log);
  Target exp_from_bits(Target base, const std::vector<BoolTarget>& bits);
  Target exp_u64(Target base, u64 exponent);
  Target exp(Target base, Target exponent, int num_bits);
  Target inverse(Target x);
  Target select(BoolTarget b, Target x, Target y);
  Target _if(BoolTarget b, Target x, Target y) { return select(b, x, y); }

  // ---- extension arithmetic (upstream gadgets/arithmetic_extension.rs) ----
  Ext mul_extension(Ext a, Ext b);
  Ext arithmetic_extension(u64 c0, u64 c1, Ext m0, Ext m1, Ext addend);
  Ext constant_extension(gl::E2 c) { return Ext{constant(c.a), constant(c.b)}; }
  Ext zero_extension() { return Ext{zero(), zero()}; }
  Ext one_extension() { return Ext{one(), zero()}; }
  Ext convert_to_ext(Target t) { return Ext{t, zero()}; }
  Ext add_virtual_extension_target() { Target a = add_virtual_target(); Target b = add_virtual_target(); return Ext{a, b}; }
  std::optional<gl::E2> target_as_constant_ext(Ext e) const;
  Ext add_extension(Ext a, Ext b) { return arithmetic_extension(1, 1, one_extension(), a, b); }
  Ext sub_extension(Ext a, Ext b) { return arithmetic_extension(1, gl::P - 1, one_extension(), a, b); }
  Ext mul_extension_with_const(u64 c, Ext a, Ext b) { return arithmetic_extension(c, 0, a, b, zero_extension()); }
  Ext mul_add_extension(Ext a, Ext b, Ext c) { return arithmetic_extension(1, 1, a, b, c); }
  Ext mul_sub_extension(Ext a, Ext b, Ext c) { return arithmetic_extension(1, gl::P - 1, a, b, c); }
  Ext mul_const_extension(u64 c, Ext x) { return mul_extension_with_const(c, one_extension(), x); }
  Ext mul_const_add_extension(u64 c, Ext x, Ext y) { return arithmetic_extension(c, 1, one_extension(), x, y); }
  Ext scalar_mul_ext(Target a, Ext b) { return mul_extension(convert_to_ext(a), b); }
  Ext square_extension(Ext a) { return mul_extension(a, a); }
  Ext add_many_extension(const std::vector<Ext>& terms);
  Ext mul_many_extension(const std::vector<Ext>& terms);
  Ext exp_u64_extension(Ext base, u64 exponent);
  Ext exp_power_of_2_extension(Ext base, int power_log);
  Ext div_add_extension(Ext x, Ext y, Ext z);
  Ext div_extension(Ext x, Ext y) { return div_add_extension(x, y, zero_extension()); }
  Ext inverse_extension(Ext y) { return div_extension(one_extension(), y); }
  void connect_extension(Ext a, Ext b) { connect(a[0], b[0]); connect(a[1], b[1]); }
  Ext select_ext(BoolTarget b, Ext x, Ext y);

  // ---- upstream gadgets/random_access.rs: v[access_index] through a RandomAccessGate (16-element lists; a list of
  // one element is returned as is)
  Target random_access(Target access_index, const std::vector<Target>& v);
  Ext random_access_extension(Target access_index, const std::vector<Ext>& v);
  std::array<Target, 4> random_access_hash(Target access_index, const std::vector<std::array<Target, 4>>& v);
  // upstream gadgets/interpolation.rs `interpolate_coset`: the interpolant of (coset_shift * g^i, values[i]), i < 16,
  // evaluated at `evaluation_point`, on one CosetInterpolationGate row
  Ext interpolate_coset(Target coset_shift, const std::vector<Ext>& values, Ext evaluation_point);
  // upstream hash/poseidon.rs `mds_layer_circuit`: the MDS layer of 12 extension targets on one PoseidonMdsGate row
  std::array<Ext, 12> poseidon_mds_layer(const std::array<Ext, 12>& state);

  // ---- Poseidon (v1) in-circuit: upstream gates/poseidon.rs + hash/poseidon.rs `permute_swapped` ----
  std::array<Target, 12> poseidon_permute_swapped(const std::array<Target, 12>& in, BoolTarget swap);
  std::array<Target, 12> poseidon_permute(const std::array<Target, 12>& in) { return poseidon_permute_swapped(in, _false()); }
  // hash_n_to_hash_no_pad::<PoseidonHash> (overwrite sponge, rate 8) and hash_or_noop
  std::array<Target, 4> hash_n_to_hash_no_pad(const std::vector<Target>& inputs);
  std::array<Target, 4> hash_or_noop(const std::vector<Target>& inputs);

  // ---- split/join, range checks (gadgets/split_join.rs, split_base.rs, range_check.rs) ----
  std::vector<BoolTarget> split_le(Target integer, int num_bits);
  Target le_sum(const std::vector<BoolTarget>& bits);
  void range_check(Target x, int n_log) { split_le(x, n_log); }
  std::pair<Target, Target> split_low_high(Target x, int n_log, int num_bits);

  // ---- reference: u32 gadgets (src/common/u32/gadgets/*.rs) ----
  Target constant_u32(uint32_t c) { return constant(c); }
  std::pair<Target, Target> mul_add_u32(Target x, Target y, Target z);
  std::pair<Target, Target> add_u32(Target a, Target b) { return mul_add_u32(a, one(), b); }
  std::pair<Target, Target> mul_u32(Target a, Target b) { return mul_add_u32(a, b, zero()); }
  Target interleave_u32(Target x);
  std::pair<Target, Target> uninterleave_to_u32(Target x);
  std::pair<Target, Target> and_xor_u32_to_u32(Target x, Target y);
  Target and_u32(Target x, Target y) { return and_xor_u32_to_u32(x, y).first; }
  Target xor_u32(Target x, Target y) { return and_xor_u32_to_u32(x, y).second; }
  std::array<Target, 2> and_u64(std::array<Target, 2> x, std::array<Target, 2> y);
  std::array<Target, 2> xor_u64(std::array<Target, 2> x, std::array<Target, 2> y);
  std::array<Target, 2> lsh_u64(std::array<Target, 2> x, int n);
  std::array<Target, 2> rsh_u64(std::array<Target, 2> x, int n);

  // ---- reference: Poseidon2 gate (src/common/poseidon2/poseidon2.rs:585-609) ----
  std::array<Target, 12> poseidon2_permute_targets(const std::array<Target, 12>& in);

  // ---- build (upstream CircuitBuilder::build) ----
  Circuit build();

  std::vector<Target> input_targets;  // filled by the circuit emitter (per-proof inputs)

 private:
  struct ArithKey {
    u64 c0, c1;
    Target m0, m1, ad;
    bool operator<(const ArithKey& o) const {
      if (c0 != o.c0) return c0 < o.c0;
      if (c1 != o.c1) return c1 < o.c1;
      if (m0 != o.m0) return m0 < o.m0;
      if (m1 != o.m1) return m1 < o.m1;
      return ad < o.ad;
    }
  };
  struct ConstGen {
    int row, constant_index, wire_index;
  };
  int virtual_index_ = 0;
  std::vector<GateInstance> rows_;
  std::vector<std::pair<Target, Target>> copy_constraints_;
  std::map<u64, Target> constants_to_targets_;
  std::unordered_map<Target, u64, TargetHash> targets_to_constants_;
  std::map<ArithKey, Target> base_arithmetic_results_;
  std::map<ArithKey, Ext> ext_mul_results_;  // keyed on (c0, a0,a1 packed)...
  std::map<std::tuple<int, u64, u64, int>, std::pair<int, int>> current_slots_;
  std::vector<ConstGen> constant_generators_;
  std::vector<Generator> generators_;
  std::vector<Target> public_inputs_;
  std::map<std::array<Target, 4>, Ext> mul_ext_memo_;
  struct ExtArithKey {
    u64 c0, c1;
    std::array<Target, 6> t;
    bool operator<(const ExtArithKey& o) const {
      if (c0 != o.c0) return c0 < o.c0;
      if (c1 != o.c1) return c1 < o.c1;
      return t < o.t;
    }
  };
  std::map<ExtArithKey, Ext> ext_arithmetic_results_;
};

Generator gate_op_generator(GateKind kind, const u64 constants[2], int row, int i);
// FRI arity schedule (upstream FriReductionStrategy::ConstantArityBits)
std::vector<int> fri_reduction_arity_bits(const CircuitConfig& cfg, int degree_bits);

}  // namespace p25
