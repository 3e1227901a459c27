// Recursive verification of this library's own proofs: a plonky2 circuit that verifies
// `n_proofs` proofs of one inner circuit -- the natural consumer of a batch of plonky3-verifier proofs
// ("aggregation").  Host code, once per (inner circuit, n_proofs) shape; the resulting Circuit is proved on the GPU
// like any other.
//
// What is restated and from where:
//  * the in-circuit constraint evaluators `Gate::eval_unfiltered_circuit` of the reference's four gates
//      Poseidon2Gate           /root/reference/src/common/poseidon2/poseidon2_gate.rs:312-397 (helpers poseidon2.rs:381-500)
//      U32ArithmeticGate       src/common/u32/gates/arithmetic_u32.rs:178-245
//      U32InterleaveGate       src/common/u32/gates/interleave_u32.rs:143-189
//      UninterleaveToU32Gate   src/common/u32/gates/uninterleave_to_u32.rs:164-228
//    and of the upstream gates the inner circuit uses (Constant, PublicInput, BaseSum<2>, Arithmetic, MulExtension,
//    Exponentiation; plonky2 @ 3de92d9 gates/*.rs, absent crate, restated from their definitions);
//  * the structure of upstream's recursive verifier (plonk/recursive_verifier.rs `verify_proof`,
//    plonk/vanishing_poly.rs `eval_vanishing_poly_circuit`, iop/challenger.rs `RecursiveChallenger`,
//    hash/merkle_proofs.rs `verify_merkle_proof_to_cap_with_cap_index`, fri/recursive_verifier.rs).
//
// Gate set (round 3): cap entries and the evaluation compared at every FRI layer are selected with RandomAccessGate,
// reductions with powers of a challenge run on ReducingGate / ReducingExtensionGate, the fold of a FRI layer is one
// CosetInterpolationGate row -- as upstream's verifier does (`random_access_hash`, `ReducingFactorTarget`,
// `interpolate_coset`).
// When PoseidonGate is evaluated in-circuit (depth >= 2) every MDS layer is one PoseidonMdsGate row (`mds_layer_circuit`).
// DEVIATION (documented in DESIGN.md): upstream's in-circuit PoseidonGate evaluator runs its FAST partial rounds
// (here: the defining rounds, one PoseidonMdsGate row per round); upstream hands the RandomAccessGate's two extra
// constant wires to its constant allocator, this builder does not; and the order of the gadget calls is this file's,
// not checked against upstream's.  So the circuit uses upstream's gate set and proves
// the same statement, but it is NOT row-for-row the circuit `builder.verify_proof::<C>()` would emit
// (4,950 rows per inner fib-64 proof, 3,650 of them PoseidonGate).
#pragma once
#include <vector>
#include "builder.h"

namespace p25 {

// Constraints of one row of gate `kind`, evaluated in-circuit on extension targets (upstream
// `Gate::eval_unfiltered_circuit`): `wires` = the row's wires opened at zeta, `consts` = its two constants,
// `pih` = the public-inputs hash.  Same constraint order as the base / extension evaluators.
std::vector<Ext> eval_gate_circuit(CircuitBuilder& b, GateKind kind, const std::vector<Ext>& wires, const Ext consts[2],
                                   const std::array<Target, 4>& pih);

// Test circuit in the spirit of the reference's `test_eval_fns` (poseidon2_gate.rs:575-581): inputs = the wires and
// constants of one row as extension elements followed by the expected constraint values; the circuit evaluates the
// gate in-circuit and connects every constraint to its expectation.
Circuit build_gate_eval_circuit(GateKind kind);

// The recursive verifier: inputs = n_proofs inner proofs, each in the flat layout of include/p25.h.  `digest` /
// `cs_cap` are the inner circuit's verifier data (VerifierOnlyCircuitData), baked in as constants.
// expose_commitment: the new circuit registers FOUR public inputs -- hash_no_pad over the identifiers of the proofs it
// verifies, where a proof's identifier is its own public inputs if it has any (an aggregate further down) and
// hash_no_pad(its wires cap) otherwise (a leaf): stacked 2-to-1 aggregators thereby expose a Poseidon tree root over
// the batch (the north star's "final aggregation" needs something to show for itself).
Circuit build_recursive_verifier(const Circuit& inner, const u64 digest[4], const std::vector<u64>& cs_cap, int n_proofs,
                                 bool expose_commitment = false);

}  // namespace p25
