// ORACLE -- test infrastructure only.  Nothing in plonky2.5_amd/ may include, link or call this.
//
// The oracle's own view of a circuit: parsed from the circuit blob (format documented in
// plonky2.5_amd/csrc/circuit_io.h), with no code shared with the product.
#pragma once
#include <string>
#include <vector>
#include "ref_field.h"

enum RGate {  // numbering = blob encoding
  RG_NOOP = 0, RG_CONSTANT, RG_PUBLIC_INPUT, RG_BASE_SUM, RG_U32_INTERLEAVE, RG_U32_UNINTERLEAVE,
  RG_ARITHMETIC, RG_MUL_EXT, RG_EXPONENTIATION, RG_U32_ARITHMETIC, RG_POSEIDON2,
  RG_ARITH_EXT,   // ArithmeticExtensionGate { num_ops: 10 }   (recursion: SURVEY.md 8f-4)
  RG_POSEIDON,    // PoseidonGate (Poseidon v1 with swap)
  RG_RANDOM_ACCESS,   // RandomAccessGate { bits: 4, num_copies: 4, num_extra_constants: 2 }
  RG_REDUCING,        // ReducingGate { num_coeffs: 43 }
  RG_REDUCING_EXT,    // ReducingExtensionGate { num_coeffs: 32 }
  RG_COSET_INTERP,    // CosetInterpolationGate { subgroup_bits: 4, degree: 6, .. }
  RG_POSEIDON_MDS,    // PoseidonMdsGate
  RG_NUM
};
enum RGen {
  RGEN_CONSTANT = 0, RGEN_RANDOM, RGEN_ARITHMETIC, RGEN_MUL_EXT, RGEN_QUOTIENT_EXT, RGEN_BASE_SPLIT,
  RGEN_WIRE_SPLIT, RGEN_BASE_SUM, RGEN_LOW_HIGH, RGEN_EXPONENTIATION, RGEN_POSEIDON2,
  RGEN_U32_ARITHMETIC, RGEN_U32_INTERLEAVE, RGEN_U32_UNINTERLEAVE,
  RGEN_ARITH_EXT, RGEN_POSEIDON, RGEN_RANDOM_ACCESS, RGEN_REDUCING, RGEN_REDUCING_EXT,
  RGEN_COSET_INTERP, RGEN_POSEIDON_MDS, RGEN_NUM
};
struct RGenerator {
  u32 kind;
  u64 c0, c1;
  int aux;
  u32 n_deps, n_outs;
  size_t arg_off;  // into RCircuit::gen_args: deps then outs (target indices)
};
struct RGateType {
  u32 kind;
  int selector_index, group_start, group_end;
};
struct RCircuit {
  size_t num_random_fill = 0;  // RandomValueGenerators (upstream randomises the PublicInputGate's unused wires)
  int degree_bits, num_wires, num_routed, num_constants, num_challenges, quotient_degree_factor;
  int rate_bits, cap_height, pow_bits, num_queries, num_selectors, num_gate_constraints;
  int num_partial_products, pi_row;
  size_t num_virtual, num_inputs;
  std::vector<int> arity_bits;
  std::vector<RGateType> gates;                 // sorted by (degree, id)
  std::vector<u32> row_kind;
  std::vector<std::vector<u64>> constants_sigmas;  // selectors | constants | sigmas
  std::vector<u64> k_is;
  std::vector<u32> input_targets, rep;
  std::vector<u32> public_inputs;               // target index of every registered public input (blob field 10)
  std::vector<RGenerator> gens;
  std::vector<u32> gen_args;
  size_t n() const { return (size_t)1 << degree_bits; }
  size_t num_targets() const { return n() * num_wires + num_virtual; }
  int num_cs() const { return (int)constants_sigmas.size(); }
  int num_constants_total() const { return num_selectors + num_constants; }  // "constants" opened
};
RCircuit ref_circuit_parse(const unsigned char* blob, size_t len);

// Witness generation: upstream generate_partial_witness -- work-list of
// generators over a partition witness; copy-constraint conflicts and generators that never run
// are reported the way upstream panics.  `seed` feeds the RandomValueGenerators deterministically.
struct RWitnessResult {
  int status;             // 0 ok, 4 conflict ("set twice with different values"), 5 generators not run
  std::string message;
  std::vector<std::vector<u64>> wires;  // [num_wires][n]  (full_witness: unset wires are 0)
  std::vector<u64> public_inputs;       // upstream `partition_witness.get_targets(&prover_data.public_inputs)`
};
// keep_going: on a copy-constraint conflict keep the partition's FIRST value, finish the run and return the wires
// with status 4 (test infrastructure: such a witness satisfies every copy constraint by construction, so whatever is
// wrong with it must show up as a violated GATE constraint -- tests/test_recursion_cpu.py)
RWitnessResult ref_generate_witness(const RCircuit& c, const u64* inputs, u64 seed, const u64* filler = nullptr,
                                    bool keep_going = false);
u64 ref_random_fill(u64 seed, u64 k);
