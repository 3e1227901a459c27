// ORACLE -- test infrastructure only.  Nothing in plonky2.5_amd/ may include, link or call this.
//
// Single-purpose CPU restatement of the plonky2 prover and verifier for one circuit blob:
// upstream plonky2 @ 3de92d9 `prover::prove` / `verifier::verify` (absent third-party crate,
// Cargo.toml:15-19; reached from /root/reference/src/p3/mod.rs:260 and :266), restated from
// SURVEY.md App. A.3-A.11.  PARITY NOTE: no fixture in the reference tree pins proof bytes, so
// byte-parity with a real upstream run is unpinned; what is pinned is (a) the hash/field KATs,
// (b) the in-circuit plonky3 verifier semantics via the artifact (witness generation fails on any
// mismatch), (c) acceptance by the restated verifier, (d) CPU == GPU equality.
#pragma once
#include <string>
#include <vector>
#include "ref_circuit.h"
#include "ref_hash.h"

int ref_set_tuned(int on);                 // AVX-512 Merkle hashing for the timed baseline leg; returns what is now in effect
void ref_set_threads(int n);               // process-wide width of parallel loops
void ref_set_thread_local_threads(int n);  // override for the calling thread (0 = none)

struct RPolyBatch {
  size_t n_polys = 0;
  int log_n = 0, rate_bits = 0;
  std::vector<std::vector<u64>> coeffs;  // [n_polys][n]
  std::vector<u64> leaves_flat;          // [n << rate_bits][n_polys] row-major, bit-reversed index order
  const u64* leaf(size_t i) const { return leaves_flat.data() + i * n_polys; }
  RMerkleTree tree;
};
RPolyBatch ref_commit_values(const std::vector<std::vector<u64>>& values, int rate_bits, int cap_height);
RPolyBatch ref_commit_coeffs(const std::vector<std::vector<u64>>& coeffs, int rate_bits, int cap_height);

struct RPrecomputed {
  RPolyBatch constants_sigmas;
  RHash circuit_digest;
};
RPrecomputed ref_precompute(const RCircuit& c);

struct RProof {
  std::vector<RHash> wires_cap, zs_cap, quotient_cap;
  std::vector<RE2> constants, sigmas, wires, zs, zs_next, pps, quotient;
  std::vector<std::vector<RHash>> fri_caps;
  struct Query {
    std::vector<std::vector<u64>> initial_leaf;            // per oracle
    std::vector<std::vector<RHash>> initial_path;
    std::vector<std::vector<RE2>> step_evals;              // per FRI layer
    std::vector<std::vector<RHash>> step_path;
  };
  std::vector<Query> queries;
  std::vector<RE2> final_poly;
  u64 pow_witness = 0;
  std::vector<u64> public_inputs;   // ProofWithPublicInputs::public_inputs
};
struct RTimings {
  double witness = 0, wires_commit = 0, zs = 0, zs_commit = 0, quotient = 0, quotient_commit = 0, openings = 0,
         fri = 0, total = 0;
};
// flat u64 layout shared with the product (include/p25.h "proof layout")
size_t ref_proof_words(const RCircuit& c);
std::vector<u64> ref_proof_flatten(const RCircuit& c, const RProof& p);
RProof ref_proof_unflatten(const RCircuit& c, const u64* w);

// Stages of the prover, callable on their own (isolated parity tests of the product's p25_partial_products /
// p25_quotient / p25_fri_prove entry points).
std::vector<std::vector<u64>> ref_partial_products(const RCircuit& c, const std::vector<std::vector<u64>>& wires_values,
                                                   const std::vector<u64>& betas, const std::vector<u64>& gammas);
std::vector<std::vector<u64>> ref_quotient_chunks(const RCircuit& c, const RPolyBatch& constants_sigmas,
                                                  const RPolyBatch& wires, const RPolyBatch& zs_batch,
                                                  const std::vector<u64>& betas, const std::vector<u64>& gammas,
                                                  const std::vector<u64>& alphas, const u64* pih /*public-inputs hash [4]*/);
struct RFriParams {
  int degree_bits, rate_bits, cap_height;
  std::vector<int> arity_bits;
  int pow_bits, num_queries;
};
int ref_fri_prove(const RFriParams& fp, const std::vector<RE2>& final_poly, RChallenger& ch,
                  const RPolyBatch* const* oracles, RProof& out, std::vector<size_t>* indices_out, std::string* msg);

// returns 0 or a p25 status code (4 witness conflict, 5 generators not run, 6 opening in subgroup)
int ref_prove(const RCircuit& c, const RPrecomputed& pre, const u64* inputs, u64 seed, RProof& out,
              RTimings* tm, std::string* msg, const u64* filler = nullptr);
// returns 0 if the proof verifies, else a non-zero code with a message
int ref_verify(const RCircuit& c, const RHash& circuit_digest, const std::vector<RHash>& constants_sigmas_cap,
               const RProof& p, std::string* msg);
