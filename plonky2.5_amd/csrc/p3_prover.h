// Native plonky3 prover for the Fibonacci AIR (uni-STARK + two-adic FRI PCS + Poseidon2 Merkle MMCS +
// width-12 duplex challenger), producing the per-proof INPUT of the hot path.
//
// The reference cannot produce such proofs (the plonky3 prover lives in an external fork,
// /root/reference/README.md:13-18); it only ships one: artifacts/proof_fibonacci.json.  This prover
// follows the conventions that the reference's in-circuit verifier fixes:
//   transcript          src/p3/challenger.rs:70-169, order src/p3/verifier.rs:135-139, 258, 363-382
//   MMCS hash/compress  src/p3/commit.rs:23-60, paths commit.rs:62-129
//   domains/selectors   src/p3/serde/two_adic.rs:48-147
//   AIR + folding       src/p3/mod.rs:176-221, src/p3/air.rs:63-118, verifier.rs:199-239
//   reduced openings    src/p3/verifier.rs:296-338;  FRI fold  verifier.rs:441-516
//   proof data model    src/p3/serde/proof.rs:16-355 (flattened in `add_virtual_to` order, :357-373)
// Pinned by the artifact: for log_n = 6, 100 queries, 16 PoW bits and the artifact's PoW witness it
// must reproduce all 15,751 field elements of artifacts/proof_fibonacci.json (tests/test_p3_prover.py).
// Host code (data generator for the path's input side); not on the per-proof hot path.
#pragma once
#include <string>
#include <vector>
#include "p3_circuit.h"

namespace p25 {

struct P3ProveParams {
  int log_n = 6;             // trace height 2^log_n (the artifact: 64 rows)
  int log_blowup = 1;        // FriConfig.log_blowup (src/p3/mod.rs:242-246 uses 1); 2 / 3 hold AIRs of degree up to 5 / 9
  int num_queries = 100;
  int pow_bits = 16;
  // proof-of-work witness search starts here (plonky3 grinds with find_any, so any valid witness
  // is a legitimate proof; different witnesses give different query indices = distinct batch items)
  u64 pow_start = 0;
  int threads = 1;
};
// Returns the proof as the flat input vector (add_virtual_to order) and its shape.
std::vector<u64> p3_prove_fibonacci(const P3ProveParams& prm, P3Config& cfg_out);
// Same prover for any AIR given as a program (p3_circuit.h: AirProgram) and its trace, col[c][row];
// constraint degree <= 2^log_blowup + 1 (2^log_quotient_degree quotient chunks; the reference's proof model has one).  Throws
// std::logic_error("quotient identity ...") if the trace does not satisfy the AIR.
std::vector<u64> p3_prove_air(const AirProgram& air, const std::vector<std::vector<u64>>& col, const P3ProveParams& prm,
                              P3Config& cfg_out);
// serde-JSON text in the reference's format (proof.rs:16-19: field elements are {"value": u64})
std::string p3_inputs_to_json(const std::vector<u64>& inputs, const P3Config& cfg);

}  // namespace p25
