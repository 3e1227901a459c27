// The plonky3-verifier circuit: host-side restatement of /root/reference/src/p3 (mod.rs, verifier.rs,
// challenger.rs, commit.rs, extension.rs, air.rs, serde/{proof,two_adic}.rs).  Emits the same gates in
// the same order through the CircuitBuilder of builder.h.  Runs once per circuit shape.
#pragma once
#include <string>
#include <vector>
#include "builder.h"

namespace p25 {

// src/p3/serde/fri.rs:3-8
struct P3FriConfig {
  int log_blowup = 1;
  int num_queries = 100;
  int proof_of_work_bits = 16;
};
// src/p3/serde/proof.rs:401-410; derived from the proof's shape at src/p3/mod.rs:74-87
struct P3Config {
  P3FriConfig fri_config;
  int log_quotient_degree = 0;
  int log_trace_height = 6;
  int trace_width = 3;
  int opening_matrix_log_max_height = 7;
  int opening_proof_query_openings_opened_values_length = 2;
  int degree_bits = 6;
  // number of field elements of one proof in `add_virtual_to` order
  size_t num_inputs() const;
};

// Proof<Target> (src/p3/serde/proof.rs:349-383)
struct P3CommitPhaseStep {
  Ext sibling_value;
  std::vector<std::array<Target, 4>> opening_proof;
};
struct P3BatchOpening {
  std::vector<std::vector<Target>> opened_values;
  std::vector<std::array<Target, 4>> opening_proof;
};
struct P3ProofTarget {
  std::array<Target, 4> trace_commit, quotient_commit;
  std::vector<Ext> trace_local, trace_next;
  std::vector<std::vector<Ext>> quotient_chunks;
  std::vector<std::array<Target, 4>> commit_phase_commits;
  std::vector<std::vector<P3CommitPhaseStep>> query_proofs;
  Ext final_poly;
  Target pow_witness;
  std::vector<std::array<P3BatchOpening, 2>> query_openings;
  int degree_bits;
};

// src/p3/air.rs:20-27
struct VerifierConstraintFolder {
  std::vector<Ext> trace_local, trace_next;
  Ext is_first_row, is_last_row, is_transition, alpha, accumulator;
  // air.rs:69-88, 90-118
  void assert_zero(CircuitBuilder& cb, Ext x);
  void assert_eq(CircuitBuilder& cb, Ext x, Ext y);
  void when_assert_eq(CircuitBuilder& cb, Ext condition, Ext x, Ext y);
};
// src/p3/air.rs:10-18 (the plugin interface a user implements per inner STARK)
struct Air {
  virtual ~Air() {}
  virtual std::string name() const = 0;
  virtual int width() const = 0;
  virtual void eval(VerifierConstraintFolder& folder, CircuitBuilder& cb) const = 0;
};
// src/p3/mod.rs:160-221 (the test's FibonacciAir)
struct FibonacciAir : Air {
  std::string name() const override { return "Fibonacci"; }
  int width() const override { return 3; }
  void eval(VerifierConstraintFolder& folder, CircuitBuilder& cb) const override;
};

// CircuitBuilderP3Arithmetic::p3_verify_proof (src/p3/mod.rs:66-94).  Registers the proof's virtual
// targets as the circuit's per-proof inputs (cb.input_targets, `add_virtual_to` order).
P3ProofTarget p3_verify_proof(CircuitBuilder& cb, const P3Config& config, const Air& air);

// gadget-level entry points (src/p3/mod.rs:40-45, 96-147), exposed for the gadget tests
Target p3_constant(CircuitBuilder& cb, u64 v);
Target p3_and(CircuitBuilder& cb, Target x, Target y);
Target p3_xor(CircuitBuilder& cb, Target x, Target y);
Target p3_lsh(CircuitBuilder& cb, Target x, int n);
Target p3_rsh(CircuitBuilder& cb, Target x, int n);
Target reverse_p3(CircuitBuilder& cb, Target x);
Target reverse_p3_bits_len(CircuitBuilder& cb, Target x, int bit_len);


// Small circuits mirroring the reference's gadget tests (src/p3/mod.rs:271-494: test_p3_and / xor /
// lsh / rsh / reverse; src/p3/commit.rs:173-198: test_compress).  Inputs are virtual targets
// [operands..., expected...]; the gadget's result is connected to `expected`, so a wrong expectation
// surfaces as a witness conflict exactly like a failing `connect` upstream.
enum GadgetKind { GADGET_AND = 0, GADGET_XOR, GADGET_LSH, GADGET_RSH, GADGET_REVERSE, GADGET_COMPRESS, GADGET_EXP };
Circuit build_gadget_circuit(int kind, int param);

}  // namespace p25
