// Upstream's `CircuitData::to_bytes` / `from_bytes` for the gate set this library knows (circuit_bytes.cpp).
#pragma once
#include <vector>
#include "builder.h"
namespace p25 {
// The constants/sigmas commitment of a circuit (upstream ProverOnlyCircuitData::constants_sigmas_commitment and
// the verifier data), as host copies of the device-resident tables: computed on the GPU, never on the host.
struct CircuitCommitment {
  const u64* coeffs;   // [num_cs][n] coefficient form
  const u64* lde;      // [num_cs][n << rate_bits] values on the coset at bit-reversed index (= upstream's leaf order)
  const u64* tree;     // Merkle tree, level after level (leaf digests first), 4 words per node; the cap is the tail
  u64 digest[4];       // circuit_digest
  std::vector<Target> public_inputs;
};
std::vector<uint8_t> circuit_data_to_bytes(const Circuit& c, const CircuitCommitment& cm);
// input_target_indices: the targets the host will assign per proof (`pw.set_target` order), as target indices --
// they are not part of CircuitData.  digest_out: the circuit digest the bytes carry (the caller compares it with the
// one the device recomputes).
Circuit circuit_data_from_bytes(const uint8_t* data, size_t len, const uint32_t* input_target_indices, size_t n_inputs,
                                u64 digest_out[4]);
}  // namespace p25
