// ORACLE -- test infrastructure only.  Nothing in plonky2.5_amd/ may include, link or call this.
//
// CPU restatement of the hashes on the proving path.
//  * Poseidon (v1), width 12: the prover's own hash (`type C = PoseidonGoldilocksConfig`,
//    /root/reference/src/p3/mod.rs:229).  Defined in the absent third-party crate plonky2 @ 3de92d9
//    (Cargo.toml:15-19: hash/poseidon.rs `Poseidon::poseidon`, naive form: for each of the 30
//    rounds add the 12 round constants, apply x^7 to all lanes (full rounds 0-3, 26-29) or lane 0
//    (partial rounds 4-25), multiply by the MDS matrix circ(17,15,41,16,2,28,13,13,39,18,34,20) +
//    diag(8,0,...)).  Pinned by the four known-answer vectors in the reference tree,
//    src/common/poseidon2/poseidon2_goldilocks.rs:190-211 (tests/test_oracle_kats.py).
//  * Poseidon2, width 12: /root/reference/src/common/poseidon2/poseidon2.rs:59-91 with constants
//    from poseidon2_goldilocks.rs:10-165.  Pinned by the Merkle paths inside
//    artifacts/proof_fibonacci.json (tests/test_oracle_p3_verifier.py).
//  * Sponge / Merkle conventions of upstream hashing.rs / merkle_tree.rs.
#pragma once
#include <vector>
#include "ref_field.h"

void ref_poseidon(u64 s[12]);        // optimised partial rounds (upstream's CPU form)
void ref_poseidon_naive(u64 s[12]);
void ref_poseidon_fast_partial_inputs(u64 s[12], u64 partial_in[22]);  // fast form + its 22 partial-round S-box inputs  // the definition: 30 x {add constants, S-box, MDS}
// Poseidon (v1), naive form, with the S-box-input trace upstream's PoseidonGate stores as wires (same layout as
// the Poseidon2 gate, which was cloned from it): [0..36) full rounds 1..3, [36..58) partial rounds, [58..106) full
// rounds 26..29.  trace may be null.
void ref_poseidon_trace(u64 s[12], u64* trace);
void ref_poseidon2(u64 s[12]);
// Poseidon2 with the S-box-input trace the Poseidon2Gate stores as wires (poseidon2_gate.rs:447-523):
// trace[0..36) full rounds 1..3, [36..58) partial rounds, [58..106) full rounds 4..7.
void ref_poseidon2_trace(u64 s[12], u64 trace[106]);

struct RHash {
  u64 e[4];
};
RHash ref_hash_no_pad(const u64* in, size_t n);
RHash ref_hash_or_noop(const u64* in, size_t n);
RHash ref_two_to_one(const RHash& l, const RHash& r);

// Merkle tree over row-major leaves (upstream MerkleTree::new).  levels[0] = leaf digests,
// levels.back() = cap (2^cap_height digests).
struct RMerkleTree {
  std::vector<std::vector<RHash>> levels;
  unsigned cap_height;
  const std::vector<RHash>& cap() const { return levels.back(); }
  // siblings from the leaf up to (excluding) the cap level
  std::vector<RHash> prove(size_t leaf) const;
};
RMerkleTree ref_merkle_build(const std::vector<std::vector<u64>>& leaves, unsigned cap_height);
bool ref_merkle_verify(const std::vector<u64>& leaf, size_t index, const std::vector<RHash>& cap,
                       const std::vector<RHash>& siblings);

// plonky2 Challenger (upstream iop/challenger.rs): duplex sponge, rate 8,
// challenges popped from the END of the 8-word output buffer.
struct RChallenger {
  u64 state[12];
  std::vector<u64> in, out;
  RChallenger();
  void observe(u64 x);
  void observe_hash(const RHash& h);
  void observe_cap(const std::vector<RHash>& cap);
  void observe_ext(RE2 x);
  u64 challenge();
  RE2 ext_challenge();
  void duplex();
};
