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

// An AIR as data: the reference's plugin interface is a Rust trait whose `eval` is code
// (air.rs:10-18); across a C ABI the same information travels as an expression DAG.  Nodes refer to
// earlier nodes; a constraint is a node that must vanish on the rows selected by `when`, folded in
// order exactly as VerifierConstraintFolder does (air.rs:69-118): `always` -> assert_zero(node),
// otherwise assert_zero(selector * node).  Used by BOTH sides of the path's input: the in-circuit
// verifier (ProgramAir below) and the native plonky3 prover (p3_prover.h).
struct AirProgram {
  enum Op : uint32_t { LOCAL = 0, NEXT = 1, CONST = 2, ADD = 3, SUB = 4, MUL = 5 };
  enum When : uint32_t { ALWAYS = 0, FIRST_ROW = 1, LAST_ROW = 2, TRANSITION = 3 };
  struct Node {
    uint32_t op, a, b;  // LOCAL/NEXT: a = column; ADD/SUB/MUL: a, b = earlier node indices
    u64 value;          // CONST
  };
  struct Constraint {
    uint32_t node, when;
  };
  int width = 0;
  std::vector<Node> nodes;
  std::vector<Constraint> constraints;
  // throws std::invalid_argument on malformed programs and on constraint degree > 9 (selector included; eight quotient
  // chunks).  The reference's proof model carries exactly one chunk (serde/proof.rs:41-48, degree <= 2); more chunks are this
  // library's extension of it, and log_quotient_degree() of them must not exceed the FRI log_blowup (checked by the callers).
  void validate() const;
  int node_degree(uint32_t i) const;
  // max over the constraints of their degree with the selector's counted, and uni-stark's get_log_quotient_degree of it:
  // log2_ceil(max(degree, 2) - 1) -- 0 for degree <= 2, 1 for 3, 2 for 4..5, 3 for 6..9
  int max_constraint_degree() const;
  int log_quotient_degree() const;
  // the test AIR of src/p3/mod.rs:176-221 in this form (same constraints, same order)
  static AirProgram fibonacci();

  // Generic evaluation, nodes computed on demand in constraint order (so that the circuit form emits
  // its gates in the order the reference's hand-written `eval` does).  ops: cst(u64), add, sub, mul.
  template <class T, class Ops, class Emit>
  void fold(const std::vector<T>& local, const std::vector<T>& next, const T sel[4], Ops& ops, Emit&& emit) const {
    std::vector<T> val(nodes.size());
    std::vector<char> have(nodes.size(), 0);
    for (const Constraint& c : constraints) {
      // iterative post-order evaluation of the sub-DAG under c.node
      std::vector<uint32_t> stack{c.node};
      while (!stack.empty()) {
        uint32_t i = stack.back();
        if (have[i]) {
          stack.pop_back();
          continue;
        }
        const Node& nd = nodes[i];
        if (nd.op == LOCAL) {
          val[i] = local[nd.a];
        } else if (nd.op == NEXT) {
          val[i] = next[nd.a];
        } else if (nd.op == CONST) {
          val[i] = ops.cst(nd.value);
        } else {
          if (!have[nd.a]) {
            stack.push_back(nd.a);
            continue;
          }
          if (!have[nd.b]) {
            stack.push_back(nd.b);
            continue;
          }
          val[i] = nd.op == ADD ? ops.add(val[nd.a], val[nd.b]) : nd.op == SUB ? ops.sub(val[nd.a], val[nd.b]) : ops.mul(val[nd.a], val[nd.b]);
        }
        have[i] = 1;
        stack.pop_back();
      }
      emit(c.when == ALWAYS ? val[c.node] : ops.mul(sel[c.when], val[c.node]));
    }
  }
};
struct ProgramAir : Air {
  AirProgram prog;
  explicit ProgramAir(AirProgram p) : prog(std::move(p)) { prog.validate(); }
  std::string name() const override { return "Program"; }
  int width() const override { return prog.width; }
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
enum GadgetKind { GADGET_AND = 0, GADGET_XOR, GADGET_LSH, GADGET_RSH, GADGET_REVERSE, GADGET_COMPRESS, GADGET_EXP, GADGET_HASH_SLICES,
                  GADGET_CONNECTED_INPUTS, GADGET_EXT_ARITH, GADGET_POSEIDON_MERKLE, GADGET_PUBLIC_INPUTS,
                  GADGET_INTERLEAVE_U32, GADGET_UNINTERLEAVE_TO_U32, GADGET_REFERENCE_GATES };
Circuit build_gadget_circuit(int kind, int param);

}  // namespace p25
