/* p25.h -- C ABI of libp25: an MI355X-native (gfx950) batch prover for the plonky2 circuit that
 * verifies a plonky3 STARK proof (the hot path of QEDProtocol/plonky2.5:
 * `data.prove(pw)` at src/p3/mod.rs:260 of the reference).
 *
 * The reference has no FFI: its boundary is the Rust call
 *     let data = builder.build::<C>();  let proof = data.prove(pw)?;      (src/p3/mod.rs:250,260)
 * against the upstream crate plonky2 @ 3de92d9 (Cargo.toml:15-19).  This header is the C boundary a
 * Rust host would bind with `extern "C"` to replace that call (INTEGRATION.md shows the binding).
 * Every entry point cites the reference interface it replaces.
 *
 * Conventions: all field elements are canonical Goldilocks u64 (< 2^64 - 2^32 + 1); caller owns all
 * buffers; calls are synchronous on return; no exceptions cross the ABI (non-zero status +
 * p25_last_error()); the library has no CPU fallback -- without a HIP device every compute entry
 * point returns P25_ERR_NO_DEVICE.
 */
#ifndef P25_H
#define P25_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t p25_status;
enum {
  P25_OK = 0,
  P25_ERR_INVALID_ARG = 1,
  P25_ERR_NO_DEVICE = 2,
  P25_ERR_HIP = 3,
  /* per-proof statuses: the upstream prover panics / returns Err in these cases */
  P25_ERR_WITNESS_CONFLICT = 4,     /* "Partition containing .. was set twice with different values" */
  P25_ERR_GENERATORS_NOT_RUN = 5,   /* "N generators weren't run" */
  P25_ERR_OPENING_IN_SUBGROUP = 6,  /* Err("Opening point is in the subgroup.") */
  P25_ERR_INTERNAL = 7,
  P25_ERR_PARSE = 8,
  /* p25_device_init_ex only: the device IS selected and usable, but the hardware-queue request probably came too late to have
   * an effect (below).  The one status that is a warning, not a failure. */
  P25_WARN_HW_QUEUES_LATE = 9,
  /* p25_comm_* / p25_gather_proofs: librccl could not be loaded, or an RCCL call failed (p25_last_error has its text) */
  P25_ERR_RCCL = 10
};

/* Last error message of the calling thread ("" if none). */
const char* p25_last_error(void);
/* Library version string. */
const char* p25_version(void);
/* Select the HIP device used by this PROCESS (one process per GPU); call once, before creating circuits AND BEFORE ANYTHING
 * ELSE IN THE PROCESS TOUCHES HIP -- required for full throughput (the hardware-queue request below is read when the HIP
 * runtime initialises; a host that must touch HIP first exports GPU_MAX_HW_QUEUES=24 itself).
 * The index is recorded and re-applied (hipSetDevice is per host thread) at every entry point, so calls from
 * any host thread run on this device.  Without it the first call adopts the thread's current device and applies the same
 * default hardware-queue request.  P25_ERR_NO_DEVICE if none. */
p25_status p25_device_init(int device_index);     /* = p25_device_init_ex(device_index, P25_DEFAULT_HW_QUEUES), warning dropped */
/* Same with the number of hardware queues the HIP runtime may spread its streams over made explicit.  The library keeps
 * 16 proofs in flight on 16 + 2 streams; ROCclr multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
 * streams sharing a queue run in order (4 -> 16 queues: 78.8 -> 91.7 proofs/s).  hw_queues > 0: the variable is set for this
 * process -- here, in this call, before the library's first HIP call, and only if the host has not exported it already; it
 * has an effect only if HIP has not been initialised yet (a host that touches HIP first exports it itself).  hw_queues = 0:
 * the environment is left alone.  Loading the library changes nothing in the process (rounds 1-4 set the variable from a
 * load-time constructor). */
#define P25_DEFAULT_HW_QUEUES 24
/* Returns P25_WARN_HW_QUEUES_LATE (device selected, library usable, p25_last_error() = the explanation) when hw_queues > 0, the
 * host had not exported GPU_MAX_HW_QUEUES, and this process already had the GPU driver open before the call (the host, torch
 * or a profiler's preloaded tool initialised the runtime first): the request then probably has no effect and 16 proving
 * streams share 4 hardware queues -- 10-40 % of the throughput, silently, before round 6. */
p25_status p25_device_init_ex(int device_index, int hw_queues);
/* What the process actually runs with (fill-in struct; every field also valid before p25_device_init). */
typedef struct {
  int32_t device_index;              /* -1 = not selected yet */
  int32_t hw_queues_requested;       /* what the library asked for at its first call (0 = nothing / not yet) */
  int32_t hw_queues_env;             /* GPU_MAX_HW_QUEUES as the environment holds it now (0 = absent) */
  int32_t hw_queues_host_exported;   /* 1 = the variable was the host's: the library left it alone */
  int32_t runtime_open_before_init;  /* 1 = this process held /dev/kfd before the library's first HIP call */
  int32_t hw_queues_setting_late;    /* 1 = requested, not host-exported, runtime already open: probably without effect */
  int32_t proving_streams, main_streams;   /* the process-wide stream pool: 16 + 2 */
  int32_t reserved[8];
} p25_runtime_info_t;
p25_status p25_runtime_info(p25_runtime_info_t* out);

/* ------------------------------------------------------------------------------------------
 * Primitives (host buffers; used by the parity tests).
 * ------------------------------------------------------------------------------------------ */

/* In-place Poseidon (v1) permutation of n width-12 states, states[n][12].
 * Replaces upstream PoseidonPermutation::permute (selected by `type C = PoseidonGoldilocksConfig`,
 * src/p3/mod.rs:229); KATs: src/common/poseidon2/poseidon2_goldilocks.rs:190-211. */
p25_status p25_poseidon_permute(uint64_t* states, size_t n);

/* In-place Poseidon2 permutation, states[n][12].
 * Replaces `Poseidon2::poseidon2` (src/common/poseidon2/poseidon2.rs:59-91). */
p25_status p25_poseidon2_permute(uint64_t* states, size_t n);

/* Merkle commitment of n_leaves leaves of `width` words, given COLUMN-major
 * (leaves_cm[c * n_leaves + l] = word c of leaf l; n_leaves a power of two >= 2^cap_height).
 * cap_out[2^cap_height][4]; tree_out (nullable) receives all levels, leaf digests first
 * (p25_merkle_tree_words words).  Replaces upstream MerkleTree::<F, PoseidonHash>::new(leaves, cap_height). */
p25_status p25_merkle_commit(const uint64_t* leaves_cm, size_t n_leaves, size_t width,
                             unsigned cap_height, uint64_t* cap_out, uint64_t* tree_out);
size_t p25_merkle_tree_words(size_t n_leaves, unsigned cap_height);

/* Polynomial-batch commitment.  polys[n_polys][2^log_n] are values on the subgroup in natural order
 * (from_coeffs = 0) or coefficients (from_coeffs = 1).  Outputs (each nullable):
 *   coeffs_out[n_polys][2^log_n]               coefficients
 *   lde_out[n_polys][2^(log_n+rate_bits)]      LDE on the coset 7*<w>, stored at BIT-REVERSED index
 *                                              (lde_out[p][rev(i)] = f_p(7 w^i)) = Merkle leaf order
 *   cap_out[2^cap_height][4]                   Merkle cap over leaves (lde_out[0..n_polys][l])_l
 * Replaces upstream PolynomialBatch::from_values / from_coeffs (blinding = false). */
p25_status p25_lde_commit(const uint64_t* polys, unsigned log_n, size_t n_polys, int from_coeffs,
                          unsigned rate_bits, unsigned cap_height, uint64_t* coeffs_out,
                          uint64_t* lde_out, uint64_t* cap_out);

/* Device-resident variants for benchmarking: all pointers are HIP device addresses owned by the
 * caller, `stream` is a hipStream_t (NULL = default stream).  Asynchronous: the caller synchronises.
 * d_tree must hold p25_merkle_tree_words(n_leaves, cap_height) words; the cap is its last
 * 4 * 2^cap_height words.  d_tmp must hold n_polys * 2^log_n words. */
p25_status p25_merkle_commit_dev(const uint64_t* d_leaves_cm, size_t col_stride, size_t n_leaves,
                                 size_t width, unsigned cap_height, uint64_t* d_tree, void* stream);
p25_status p25_lde_commit_dev(const uint64_t* d_polys, unsigned log_n, size_t n_polys, int from_coeffs,
                              unsigned rate_bits, unsigned cap_height, uint64_t* d_coeffs,
                              uint64_t* d_tmp, uint64_t* d_lde, uint64_t* d_tree, void* stream);
p25_status p25_poseidon_permute_dev(uint64_t* d_states, size_t n, void* stream);


/* ------------------------------------------------------------------------------------------
 * Circuits.  A p25_circuit is the built plonky2 circuit (upstream `CircuitData`): immutable after
 * creation, reusable for any number of proofs (`prove` borrows it immutably upstream too).
 * Building / import / export / info are host-only and work without a GPU; proving needs one.
 * ------------------------------------------------------------------------------------------ */
typedef struct p25_circuit p25_circuit;

/* Shape of the plonky3 proof being verified in-circuit: the reference's FriConfig
 * (src/p3/serde/fri.rs:3-8, values at src/p3/mod.rs:242-246) + P3Config derived from the proof's
 * shape (src/p3/mod.rs:74-87, src/p3/serde/proof.rs:401-410). */
typedef struct {
  int32_t log_blowup, num_queries, proof_of_work_bits;
  int32_t log_quotient_degree, log_trace_height, trace_width;
  int32_t opening_matrix_log_max_height, quotient_opened_len, degree_bits;
} p25_p3_config;

enum { P25_AIR_FIBONACCI = 0 }; /* the test AIR of src/p3/mod.rs:160-221 */

/* Replaces: CircuitBuilder::new(CircuitConfig::standard_recursion_config());
 *           builder.p3_verify_proof::<PoseidonHash>(proof, &air, fri_config);
 *           builder.build::<PoseidonGoldilocksConfig>()          (src/p3/mod.rs:231-250). */
p25_status p25_circuit_build_p3_verifier(const p25_p3_config* cfg, int32_t air, p25_circuit** out);
/* An AIR as data.  The reference's plugin interface for the inner STARK is the Rust
 * trait `Air` (src/p3/air.rs:10-18) whose `eval` body is code; across the C ABI the same information is
 * an expression DAG.  node.op: 0 LOCAL(column a of the current row)  1 NEXT(column a of the next row)
 * 2 CONST(value)  3 ADD(a,b)  4 SUB(a,b)  5 MUL(a,b), a/b = indices of EARLIER nodes.  A constraint is a
 * node that must vanish on the rows `when` selects (0 always, 1 first row, 2 last row, 3 transition),
 * folded in order like VerifierConstraintFolder (air.rs:69-118).  Degree 4-5 / 6-9: FOUR / EIGHT chunks with log_blowup >= 2 / 3
 * (round 6, p25_p3_prove_air_ex).  Degree (selector included) <= 2: ONE quotient chunk,
 * the reference's proof model (serde/proof.rs:41-48 `(0..1)`), p25_p3_config.log_quotient_degree = 0.  Degree 3: TWO chunks
 * (log_quotient_degree = 1; round 5): the reference's verifier (verifier.rs:115-221) and its shape derivation (mod.rs:76)
 * handle any power of two, only that `(0..1)` fixes the count -- the flat input then carries the second chunk's two
 * openings right behind the first's, and every query's quotient batch holds one row per chunk.  The FibonacciAir of
 * src/p3/mod.rs:176-221 in this form builds the very same circuit (same digest) as P25_AIR_FIBONACCI. */
typedef struct {
  uint32_t op, a, b, reserved;
  uint64_t value;
} p25_air_node;
typedef struct {
  uint32_t node, when;
} p25_air_constraint;
typedef struct {
  uint32_t width, n_nodes, n_constraints, reserved;
  const p25_air_node* nodes;
  const p25_air_constraint* constraints;
} p25_air;
/* p3_verify_proof for a user AIR: `builder.p3_verify_proof::<H>(proof, &air, fri_config)` with any `impl Air`
 * (src/p3/mod.rs:66-94, 239-250).  cfg->trace_width must equal air->width. */
p25_status p25_circuit_build_p3_verifier_air(const p25_p3_config* cfg, const p25_air* air, p25_circuit** out);
/* Native plonky3 prover (see p25_p3_prove_fibonacci) for a user AIR and its trace, trace[row * width + col],
 * 2^log_n rows.  P25_ERR_INVALID_ARG if the trace does not satisfy the AIR. */
p25_status p25_p3_prove_air(const p25_air* air, const uint64_t* trace, int32_t log_n, int32_t num_queries,
                            int32_t pow_bits, uint64_t pow_start, int32_t threads, uint64_t* inputs_out, size_t cap,
                            size_t* n_out, p25_p3_config* cfg_out);
/* The same with FriConfig.log_blowup explicit (src/p3/mod.rs:242-246; the reference's verifier reads config.log_blowup
 * generically: verifier.rs:264, 299, 378, 397).  log_blowup 1..4; 1 = p25_p3_prove_air = the reference's artifact.  log_blowup 2 / 3:
 * the LDE domain is 7*H_{4n} / 7*H_{8n}, which holds the quotient domain of AIRs of constraint degree up to 5 / 9 (selector
 * included): 2^log_quotient_degree = 4 / 8 quotient chunks, log_quotient_degree = log2_ceil(degree - 1) as in uni-stark.  The
 * flat input then carries every chunk's two openings in order, one matrix per chunk in every query's quotient batch, input
 * Merkle paths of log_n + log_blowup digests and commit-phase paths of log_n + log_blowup - 1 - i (the reference's data model,
 * serde/proof.rs:204-205, writes log_n - i: the log_blowup-1 case).  P25_ERR_INVALID_ARG if the AIR's degree needs more chunks
 * than log_blowup holds. */
p25_status p25_p3_prove_air_ex(const p25_air* air, const uint64_t* trace, int32_t log_n, int32_t log_blowup, int32_t num_queries,
                               int32_t pow_bits, uint64_t pow_start, int32_t threads, uint64_t* inputs_out, size_t cap,
                               size_t* n_out, p25_p3_config* cfg_out);

/* Small circuits mirroring the reference's gadget tests (src/p3/mod.rs:271-494 test_p3_and / xor / lsh /
 * rsh / reverse, src/p3/commit.rs:173-198 test_compress): kind 0 and(x,y) 1 xor(x,y) 2 lsh(x,param)
 * 3 rsh(x,param) 4 reverse_bits_len(x,param) 5 Poseidon2 compress(l[4],r[4]) 6 7*w_param^e with inverse
 * 7 MerkleTreeMmcs::hash_iter_slices over `param` slices of 4 words (src/p3/commit.rs:23-46, test :143-171)
 * 8 inputs a, b under one copy constraint (`connect(a, b)`) and a*b: a != b fails with
 *   P25_ERR_WITNESS_CONFLICT like upstream's "was set twice with different values"
 * 9 extension-field gadget chain (ArithmeticExtensionGate, MulExtensionGate, virtual inverse): inputs a, b, c in
 *   F_p^2 and the expected result of ((5*((a*b+c)*a-b)+(3+9X))^7 / c + a + b + a0*b
 * 10 Poseidon (v1) in-circuit: hash_or_noop of `param` words, then one Merkle step `permute_swapped([h, sibling, 0],
 *   bit)` on a PoseidonGate; inputs = leaf words, sibling[4], bit, expected parent[4].
 * 11 public inputs (upstream `register_public_input(s)`): `param` inputs x_i; the circuit registers every x_i and the
 *   running products x_0 x_1, x_0 x_1 x_2, ... as public inputs (2 * param - 1 in all).
 * 12 interleave_u32 (the reference's test_interleave_u32, src/common/u32/gadgets/interleaved_u32.rs:354-382): param 1 =
 *   that test's circuit as written (x = constant_u32(0xFFFFFFFC), no witness inputs), param 0 = x is the one input; the
 *   interleaved value is the circuit's public input (the test expects 0x5555555555555550).
 * 13 uninterleave_to_u32 (test_uninterleave_to_u32, :388-417): param 1 = x = constant(0xF555555555555555), param 0 = x is
 *   the input; public inputs = (evens, odds) (the test expects 0xC0000000, 0xFFFFFFFF).
 * 14 the reference's four gates in one circuit: inputs x, y, z (u32 values) and the expected x*y, interleave(x),
 *   interleave(y), the (evens, odds) of uninterleave_to_u32(interleave(x)), the (low, high) of x*y+z, and the first four
 *   words of Poseidon2 over those eleven values + 0 -- the circuit tests/blob_writer.py builds independently (8 gate
 *   types in two selector groups).
 * Inputs = operands followed by the expected result(s); a wrong expectation fails the proof with
 * P25_ERR_WITNESS_CONFLICT, as the failing `connect` panics upstream. */
p25_status p25_circuit_build_gadget(int32_t kind, int32_t param, p25_circuit** out);
/* ------------------------------------------------------------------------------------------
 * Recursion: verifying this library's proofs inside another plonky2 circuit.
 * ------------------------------------------------------------------------------------------ */
/* A circuit that verifies `n_proofs` (1..16) proofs of `inner` -- upstream `builder.verify_proof::<C>(proof, vd, cd)`
 * for each: Fiat-Shamir challenges on Poseidon gates, the constraint check vanishing(zeta) = Z_H(zeta) t(zeta)
 * with every gate's `eval_unfiltered_circuit` (the reference's: poseidon2_gate.rs:312-397,
 * arithmetic_u32.rs:178-245, interleave_u32.rs:143-189, uninterleave_to_u32.rs:164-228), and the FRI verifier
 * (PoW, Merkle paths, folding; cap entries and the evaluation checked at every FRI layer are selected with
 * RandomAccessGate, reductions with powers of alpha run on ReducingGate / ReducingExtensionGate, a layer's fold is
 * one CosetInterpolationGate row, as in upstream's verifier).
 * Inputs of the new circuit = the inner proofs' words in the flat layout below, concatenated.  digest4 / cs_cap: the inner circuit's verifier data, baked in as constants; pass NULL to have them
 * computed on the GPU (p25_circuit_digest).  Gate set: upstream-standard gates only, but not row-for-row upstream's
 * circuit (plonky2.5_amd/csrc/recursion.h).  A false inner proof fails with P25_ERR_WITNESS_CONFLICT. */
p25_status p25_circuit_build_recursive_verifier(p25_circuit* inner, const uint64_t* digest4, const uint64_t* cs_cap,
                                                int32_t n_proofs, p25_circuit** out);
/* Gate-level test circuit in the spirit of the reference's `test_eval_fns` (poseidon2_gate.rs:575-581): inputs = one
 * row's 135 wires and 2 constants as extension elements (c0, c1 each), the 4-word public-inputs hash, then the
 * expected value of every constraint as extension elements; the circuit evaluates gate `kind` in-circuit
 * (`eval_unfiltered_circuit`) and connects each constraint to its expectation.  kind: 1 Constant 2 PublicInput
 * 3 BaseSum 4 U32Interleave 5 UninterleaveToU32 6 Arithmetic 7 MulExtension 8 Exponentiation 9 U32Arithmetic
 * 10 Poseidon2 11 ArithmeticExtension 12 Poseidon 13 RandomAccess 14 Reducing 15 ReducingExtension
 * 16 CosetInterpolation 17 PoseidonMds (the `kind` numbering of the circuit blob). */
p25_status p25_circuit_build_gate_eval(int32_t kind, p25_circuit** out);
/* p25_circuit_build_recursive_verifier whose circuit also REGISTERS FOUR PUBLIC INPUTS (upstream
 * `builder.register_public_inputs`): hash_no_pad over the identifiers of the proofs it verifies, a proof's identifier
 * being its own public inputs when it has any (an aggregate further down the tree) and hash_no_pad(its wires cap)
 * otherwise (a leaf).  Stacked aggregators expose the root of a Poseidon tree over the batch: the "final
 * aggregation step" of BASELINE.json leaves one proof whose public inputs commit to every leaf proof.  n_proofs is free:
 * the circuit's size follows it -- over fib-64 verifier proofs 13 children still fit 2^16 rows (62,753; 8 use 38,687,
 * 14 need 2^17), which is what plonky25_amd.aggregate.widest_arity finds and bench.py uses. */
p25_status p25_circuit_build_aggregator(p25_circuit* inner, const uint64_t* digest4, const uint64_t* cs_cap,
                                        int32_t n_proofs, p25_circuit** out);

/* Circuit blob (format: plonky2.5_amd/csrc/circuit_io.h): persist a built circuit / hand it to
 * another process.  export: pass buf = NULL to query the size. */
p25_status p25_circuit_export(const p25_circuit* c, uint8_t* buf, size_t cap, size_t* len_out);
p25_status p25_circuit_import(const uint8_t* blob, size_t len, p25_circuit** out);
/* Upstream's own binary form of a built circuit: `CircuitData::to_bytes(&gate_serializer, &generator_serializer)` /
 * `CircuitData::from_bytes` (plonky2 util/serialization, restated -- the crate is absent, unpinned like the proof
 * formats; the reference's gate / generator payloads are its own `serialize` bodies: poseidon2_gate.rs:399-405,
 * 529-539, arithmetic_u32.rs:287-300, 445-464, interleave_u32.rs:237-247, 340-360, uninterleave_to_u32.rs:272-283,
 * 396-412).  Gate / generator tags = upstream's default serializer lists followed by the reference's types
 * (INTEGRATION.md section 5a gives the `impl_gate_serializer!` / `impl_generator_serializer!` declarations).
 * to_bytes needs the GPU (the constants/sigmas commitment -- LDE leaves, Merkle digests, cap, circuit digest -- is
 * part of the data: ~600 MB for the fib-64 circuit); *bytes_out is malloc'ed, release it with p25_free.
 * from_bytes: input_targets[n_inputs] = target indices of the per-proof inputs in the order the host passes their
 * values (they are not part of CircuitData; p25_circuit_input_targets returns them for a circuit built here);
 * digest4_out (nullable) receives the circuit digest stored in the bytes -- p25_circuit_digest recomputes it. */
p25_status p25_circuit_to_bytes(p25_circuit* c, uint8_t** bytes_out, size_t* len_out);
p25_status p25_circuit_from_bytes(const uint8_t* bytes, size_t len, const uint32_t* input_targets, size_t n_inputs,
                                  uint64_t* digest4_out, p25_circuit** out);
p25_status p25_circuit_input_targets(const p25_circuit* c, uint32_t* targets_out, size_t cap, size_t* n_out);
void p25_free(void* p);
void p25_circuit_destroy(p25_circuit* c);

typedef struct {
  uint64_t degree_bits, num_rows_used, num_wires, num_routed_wires, num_inputs, num_generators;
  uint64_t num_gate_types, num_selectors, num_constants_sigmas, num_gate_constraints;
  uint64_t proof_words, witness_levels, witness_slots, num_random_fill;
  /* sizes of the stage entry points' outputs: p25_partial_products writes num_challenges * (1 + num_partial_products)
   * rows, p25_quotient num_challenges * quotient_degree_factor rows, of 2^degree_bits words each */
  uint64_t num_challenges, num_partial_products, quotient_degree_factor;
  uint64_t num_public_inputs;   /* registered public inputs: the last num_public_inputs words of a flat proof */
} p25_circuit_info_t;
p25_status p25_circuit_info(p25_circuit* c, p25_circuit_info_t* out);
/* Rows per gate type, in sorted-gate order; ids_out receives up to cap gate-id strings joined by '\n'. */
p25_status p25_circuit_gate_counts(const p25_circuit* c, uint64_t* counts_out, size_t cap, char* ids_out, size_t ids_cap);
/* Verifier data (upstream VerifierOnlyCircuitData): circuit_digest[4] and the constants/sigmas cap
 * [2^cap_height][4].  Computed on the GPU at first use. */
p25_status p25_circuit_digest(p25_circuit* c, uint64_t* digest4, uint64_t* constants_sigmas_cap);

/* ------------------------------------------------------------------------------------------
 * Proving.  Replaces `data.prove(pw)` (src/p3/mod.rs:260) for a batch of independent witnesses.
 *
 *   inputs[n_proofs][num_inputs]  the plonky3 proof's field elements in `add_virtual_to` order
 *                                 (src/p3/serde/proof.rs:357-373) = what set_witness assigns (:374-383)
 *   seeds[n_proofs] (nullable)    upstream fills the PublicInputGate's 131 unused wires from the OS RNG
 *                                 (RandomValueGenerator), so real proofs are not reproducible; here the
 *                                 filler is SplitMix64(seed, wire) and proofs are deterministic.
 *                                 NULL = seed i for proof i.
 *   proofs_out[n_proofs][proof_stride_words]   flat proofs (layout below), stride >= proof_words
 *   per_proof_status[n_proofs]    P25_OK or the upstream failure mode (P25_ERR_WITNESS_CONFLICT ...);
 *                                 a bad witness fails that proof only.
 *   timings (nullable)            device milliseconds per upstream phase, summed over the batch.
 *
 * Proof layout (u64 words; E = extension element as (c0, c1); H = 4-word hash; CAP = 2^cap_height H):
 *   wires_cap CAP | plonk_zs_partial_products_cap CAP | quotient_polys_cap CAP |
 *   openings: constants E[5] | plonk_sigmas E[80] | wires E[135] | plonk_zs E[2] | plonk_zs_next E[2] |
 *             partial_products E[18] | quotient_polys E[16] |
 *   commit_phase_merkle_caps CAP[3] |
 *   query_round_proofs[28]: for each of the 4 oracles {leaf row u64[width], siblings H[15]};
 *                           for each FRI layer {evals E[16], siblings H[11,7,3]} |
 *   final_poly E[16] | pow_witness u64 | public_inputs u64[num_public_inputs]
 *                                                (sizes shown for the fib-64 circuit, which has no public inputs;
 *                                                 total 19,861 words)
 * Public inputs (upstream ProofWithPublicInputs::public_inputs): the values of the targets the circuit registered,
 * read from the witness; their Poseidon hash_no_pad is what the PublicInputGate row holds, what every gate
 * evaluator receives as `public_inputs_hash`, and what the transcript absorbs after the circuit digest.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  float witness_ms, wires_commit_ms, partial_products_ms, zs_commit_ms, quotient_ms, quotient_commit_ms,
      openings_ms, fri_ms, total_ms;
} p25_timings;
/* Threading: like upstream's `prove(&self)`, every entry point may be called from any host thread, also concurrently
 * on ONE p25_circuit (a Rust host with a rayon pool): a circuit owns its per-proof contexts, so such
 * calls are serialised inside the library (one mutex per circuit; use the batch forms, or one circuit per thread,
 * for parallelism).  Different circuits do not contend for a lock while proving (two exceptions: building a recursive
 * verifier / aggregator holds the INNER circuit's lock for the duration of the build, and p25_circuit_wait_mark takes both
 * circuits' locks); their proofs share the device through ONE pool of 16
 * proving streams + 2 main streams per process (a stream set per circuit oversubscribes the hardware queues as soon as
 * two circuits are alive: DESIGN.md section 3, docs/HISTORY.md section 3).  With `timings` != NULL, or a batch of one, the proofs run one
 * at a time with latency-oriented kernel forms; otherwise up to 16 proofs are in flight (p25_circuit_set_streams). */
p25_status p25_prove_batch(p25_circuit* c, const uint64_t* inputs, size_t n_proofs, const uint64_t* seeds,
                           uint64_t* proofs_out, size_t proof_stride_words, p25_status* per_proof_status,
                           p25_timings* timings);
/* p25_prove_batch with EXPLICIT values for the RandomValueGenerator wires instead of seeds: filler[n_proofs][num_random_fill]
 * (p25_circuit_info_t.num_random_fill = 131 for these circuits; order = PublicInputGate wires 4..134).  With the values a
 * real upstream run drew from the OS RNG, the proof is the one that run produced (tests/test_upstream_golden.py). */
p25_status p25_prove_batch_filler(p25_circuit* c, const uint64_t* inputs, size_t n_proofs, const uint64_t* filler,
                                  uint64_t* proofs_out, size_t proof_stride_words, p25_status* per_proof_status);
/* Same with every buffer resident in HBM (device pointers; d_status is uint32_t[n_proofs]).
 * Enqueues on the circuit's stream and returns; p25_circuit_sync waits. */
p25_status p25_prove_batch_dev(p25_circuit* c, const uint64_t* d_inputs, size_t n_proofs, const uint64_t* d_seeds,
                               uint64_t* d_proofs, size_t proof_stride_words, uint32_t* d_status,
                               p25_timings* timings);
/* The same over WINDOWS of one device buffer: proof i reads its num_inputs words at
 *   d_buffer + min(i * window_stride_words, last_window_offset_words),
 * i.e. equally spaced windows with the LAST one right-aligned.  This is how a level of an aggregation tree proves straight on
 * the buffer the level below wrote its proofs into, in ONE batch: an aggregator of k children takes k consecutive flat proofs
 * (window stride = k * child proof words = its num_inputs); when k does not divide the number of children n, the last of
 * the ceil(n / k) groups is the last k children -- offset (n - k) * child proof words -- and overlaps its neighbour.
 * THE COMMITMENT RULE a consumer of the root's public inputs must reproduce (plonky25_amd.aggregate.level_plan /
 * group_bounds / expected_commitment are its executable form): a level of n children folded at most `arity` at a time has
 * G = ceil(n / arity) groups of k = ceil(n / G) children; group g < G - 1 covers children [g k, (g + 1) k), group G - 1
 * covers [n - k, n); an aggregate's four public inputs are hash_no_pad over its children's identifiers in that order (a
 * child's identifier = its own public inputs if it has any, hash_no_pad(its wires cap) otherwise); levels repeat until one
 * proof is left; with N ranks every rank folds its own shard this way and one N-to-1 aggregate over the shard roots (rank
 * order) is the final proof.  Children in an overlap are verified twice and appear twice under the root. */
/* Every window holds num_inputs words (p25_circuit_info_t): d_buffer must be readable up to last_window_offset_words +
 * num_inputs; a stride below num_inputs makes consecutive windows overlap, which is legal.  window_stride_words >= 1 and
 * n_proofs * window_stride_words < 2^60, P25_ERR_INVALID_ARG otherwise. */
p25_status p25_prove_batch_dev_windows(p25_circuit* c, const uint64_t* d_buffer, size_t window_stride_words,
                                       size_t last_window_offset_words, size_t n_proofs, const uint64_t* d_seeds,
                                       uint64_t* d_proofs, size_t proof_stride_words, uint32_t* d_status);
p25_status p25_circuit_sync(p25_circuit* c);
/* Device-side ordering between the circuit's proving streams and a stream of the caller's (a hipStream_t; NULL = the
 * legacy default stream), with NO host synchronisation -- what a host needs to consume step k's proofs (copy them out,
 * hand them to RCCL: the "final aggregation" of north_star, bench.py) underneath step k+1's proving:
 *   p25_circuit_stream_join(c, s):  s waits for every proof the circuit has been asked for so far
 *   p25_circuit_wait_stream(c, s):  proofs requested from now on start only after what s holds now (e.g. the gather
 *                                   that still reads the buffer the next p25_prove_batch_dev overwrites)
 * The reference has no counterpart (data.prove(pw), src/p3/mod.rs:260, is synchronous); these replace the
 * host-blocking p25_circuit_sync between pipelined steps. */
p25_status p25_circuit_stream_join(p25_circuit* c, void* stream);
p25_status p25_circuit_wait_stream(p25_circuit* c, void* stream);
/* The same between two circuits, through events only -- chaining provers on the device, e.g. an aggregation circuit
 * (p25_circuit_build_aggregator) proving straight on the buffer its children's proofs are being written to: its inputs
 * are those flat proofs back to back, so `d_inputs` of its p25_prove_batch_dev is the producer's `d_proofs`.
 *   p25_circuit_mark(c, slot):            remember the tail of every proving stream of c under `slot` (0..P25_MAX_MARKS-1)
 *   p25_circuit_wait_mark(c, producer, slot):  what c is asked for from now on starts only after the producer's mark
 * Marks are events: recording one never blocks, and a wait issued long after the mark (plonky25_amd.aggregate.DeviceTree
 * issues a level's wait one step late) finds it already satisfied, so no hardware queue stalls on it.  Upstream's
 * counterpart is host code: `builder.verify_proof` circuits proved one after the other by `data.prove(pw)`
 * (src/p3/mod.rs:260) with the proofs passed through `PartialWitness`. */
#define P25_MAX_MARKS 16
p25_status p25_circuit_mark(p25_circuit* c, uint32_t slot);
p25_status p25_circuit_wait_mark(p25_circuit* c, p25_circuit* producer, uint32_t slot);
/* ... and a stream of the caller's waiting for a mark: p25_circuit_stream_join that can be issued late (bench.py marks
 * step k when it has enqueued it and lets the gather's side stream wait for that mark only after step k+1 has been
 * enqueued, so the wait never sits unsatisfied at the head of a hardware queue shared with a proving stream). */
p25_status p25_circuit_stream_wait_mark(p25_circuit* c, uint32_t slot, void* stream);
/* Proofs kept in flight by the batch entry points: one per-proof working set (~1.6 GB for the fib-64 circuit) and one
 * stream of the process-wide pool each; 1..32, default 16 (12 .. 20 measure the same, 24 and more are slower).  A library setting, not an
 * environment variable. */
p25_status p25_circuit_set_streams(p25_circuit* c, int32_t n_streams);
/* Measurement hook for bench.py's roofline line: when enabled, HIP events on the proving stream
 * bracket every launch of the dominant kernel (the Poseidon leaf sponge over the 135-column wires
 * LDE).  Returns accumulated device milliseconds and launch count; reset != 0 clears them.  The proving streams belong to
 * a process-wide pool: with more than one circuit proving at the same time the bracket also spans the other circuits'
 * kernels on that stream, so enable it while exactly one circuit is active (bench.py does). */
p25_status p25_circuit_kernel_stats(p25_circuit* c, int enable, int reset, double* ms_out, uint64_t* launches_out);
/* Measurement hook: the shader clock (Hz) under a full-chip Poseidon load, from the in-kernel cycle counter against
 * the constant-rate wall-clock counter; bench.py prices its VALU-instruction view with it instead of a nominal clock. */
p25_status p25_shader_clock_hz(double* hz_out);
/* Witness only (parity tests): wires_out[num_wires][2^degree_bits], column-major. */
p25_status p25_witness(p25_circuit* c, const uint64_t* inputs, uint64_t seed, uint64_t* wires_out,
                       p25_status* proof_status);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, every rank proves its own shard of the batch on its own replica of the circuit tables
 * (`prove(&self, ..)` borrows the circuit immutably: /root/reference/src/p3/mod.rs:260 -- nothing is exchanged inside a
 * proof), and ONE collective moves the finished proofs: the "final aggregation step" of north_star, RCCL over xGMI.
 * The reference has no counterpart (it has no multi-device code at all); these entry points are what a Rust host binds for
 * N > 1.  librccl is loaded on first use (dlopen "librccl.so.1"); P25_ERR_RCCL if it is absent.
 *
 *   rank 0:      p25_comm_unique_id(id)  -> hand the 128 bytes to every rank out of band (the launcher's channel)
 *   every rank:  p25_device_init(local_gpu); p25_comm_init(id, rank, world, &comm)
 *   every step:  p25_prove_batch_dev(c, ...);  p25_circuit_mark(c, slot);
 *                p25_gather_proofs(comm, c, slot, d_proofs, stride, d_status, counts, 0, d_all, d_all_status);
 *                   -- enqueue-only: runs on the communicator's own stream, which waits ON THE DEVICE for the mark, so the
 *                      gather of step k overlaps the proving of step k + 1 (double-buffer d_proofs; before a buffer is
 *                      proved into again: p25_circuit_wait_stream(c, p25_comm_stream(comm)))
 *   at the end:  p25_comm_sync(comm)  (host waits for the gathers), p25_comm_destroy(comm)
 * ------------------------------------------------------------------------------------------ */
typedef struct p25_comm p25_comm;
#define P25_COMM_ID_BYTES 128   /* = NCCL_UNIQUE_ID_BYTES */
p25_status p25_comm_unique_id(uint8_t* id_out /* [P25_COMM_ID_BYTES] */);
/* Collective: every rank of the job calls it with the same id.  The communicator lives on the device p25_device_init
 * selected (one rank per GPU: RCCL refuses two ranks on one device). */
p25_status p25_comm_init(const uint8_t* id, int32_t rank, int32_t world, p25_comm** out);
p25_status p25_comm_destroy(p25_comm* comm);   /* waits for the communicator's stream first */
int32_t p25_comm_rank(const p25_comm* comm);
int32_t p25_comm_world(const p25_comm* comm);
/* The hipStream_t every collective of this communicator is enqueued on (for p25_circuit_wait_stream, or the host's own
 * event / copy work behind a gather). */
void* p25_comm_stream(p25_comm* comm);
p25_status p25_comm_sync(p25_comm* comm);      /* host waits for everything enqueued on the communicator's stream */
/* Host-synchronous helpers for the timing protocol of a batch job (barrier on both sides of the timed region, MAX of the
 * per-rank elapsed time): one 8-byte ncclAllReduce on the communicator's stream. */
p25_status p25_comm_barrier(p25_comm* comm);
p25_status p25_comm_max_f64(p25_comm* comm, double* value /* in: this rank's, out: the maximum */);
/* Gather: rank q contributes counts[q] proofs, d_proofs[counts[rank]][proof_stride_words] with their statuses
 * d_status[counts[rank]] (what p25_prove_batch_dev wrote); dst_rank receives all of them in rank order -- global proof order
 * for the contiguous block partition -- into d_all_proofs[sum counts][proof_stride_words] and d_all_status[sum counts] (both
 * ignored on the other ranks, may be NULL there).  counts[world] must be the same on every rank; shards may differ in size
 * and may be empty.  Grouped ncclSend / ncclRecv (every sender has its own xGMI link to the root), the root's own block a
 * device-to-device copy.  Ordered behind `circuit` on the device: mark_slot >= 0 waits for that p25_circuit_mark, mark_slot < 0
 * for everything the circuit has been asked for so far (p25_circuit_stream_join); circuit == NULL: no wait (buffers the host
 * has synchronised itself).  Enqueue-only: returns at once; p25_comm_sync (or work on p25_comm_stream) observes completion. */
p25_status p25_gather_proofs(p25_comm* comm, p25_circuit* circuit, int32_t mark_slot, const uint64_t* d_proofs,
                             size_t proof_stride_words, const uint32_t* d_status, const size_t* counts, int32_t dst_rank,
                             uint64_t* d_all_proofs, uint32_t* d_all_status);

/* ------------------------------------------------------------------------------------------
 * Stages of the prover on their own (host buffers; the fine-grained entry points of SURVEY.md 8b, used by the
 * isolated parity tests of rows a6-a10).  Each replaces one upstream function reached from `data.prove(pw)`
 * (src/p3/mod.rs:260); NC = num_challenges (2), NP = partial products per challenge, n = 2^degree_bits.
 * ------------------------------------------------------------------------------------------ */
/* Challenger script (upstream iop/challenger.rs `Challenger<F, PoseidonHash>`: duplex sponge, rate 8, challenges
 * popped from the end of the output buffer): for segment k observe seg_len[k] words of `observe` (consumed in
 * order), then draw n_challenges[k] (<= 64) challenges, appended to challenges_out. */
p25_status p25_transcript(const uint64_t* observe, const uint32_t* seg_len, const uint32_t* n_challenges,
                          size_t n_segments, uint64_t* challenges_out);
/* upstream prover.rs `wires_permutation_partial_products_and_zs`: wires[num_wires][n] (witness values, column
 * major) + betas[NC], gammas[NC] -> out[NC * (1 + NP)][n]: the NC Z polynomials, then the NC * NP partial
 * products (values on the subgroup, natural row order). */
p25_status p25_partial_products(p25_circuit* c, const uint64_t* wires, const uint64_t* betas, const uint64_t* gammas,
                                uint64_t* out);
/* upstream prover.rs `compute_quotient_polys` (+ "split up quotient polys"), including every gate's
 * `eval_unfiltered_base_batch` (the reference's: poseidon2_gate.rs:233-310, arithmetic_u32.rs:303-366,
 * interleave_u32.rs:250-287, uninterleave_to_u32.rs:285-335): wires[num_wires][n] and zs_pp[NC*(1+NP)][n]
 * (values) + betas, gammas, alphas [NC] -> out[NC * 2^rate_bits][n]: coefficients of the quotient chunks. */
p25_status p25_quotient(p25_circuit* c, const uint64_t* wires, const uint64_t* zs_pp, const uint64_t* betas,
                        const uint64_t* gammas, const uint64_t* alphas, uint64_t* out);
/* upstream `OpeningSet::new` / `PolynomialCoeffs::eval` (a9): coeffs[n_polys][2^log_n] (base-field coefficients) evaluated at
 * the extension point (point[0] + point[1] X) * scale -> out[n_polys][2].  scale = 1 for the openings at zeta, the
 * subgroup generator for the next-row openings at g*zeta. */
p25_status p25_eval_polys(const uint64_t* coeffs, size_t n_polys, unsigned log_n, const uint64_t* point, uint64_t scale,
                          uint64_t* out);
/* upstream fri/prover.rs `fri_proof` on one batched polynomial: coeffs[2][2^log_n] (the two components of its
 * extension-field coefficients), transcript initialised by observing seed[n_seed].  Commit phase (LDE on
 * 7*<w>, 2^arity-ary leaves, Merkle caps, fold by beta), final polynomial, PoW grind, num_queries (<= 64) query rounds.
 *   out: CAP[n_layers] | betas E[n_layers] | final_poly E[2^(log_n - sum arity)] | pow_witness |
 *        query indices u64[num_queries] | per query, per layer {evals E[2^arity], siblings H[..]}
 * p25_fri_prove_words gives the size (0 for an invalid shape).  *status_out: P25_OK or P25_ERR_INTERNAL (no PoW witness). */
size_t p25_fri_prove_words(unsigned log_n, unsigned rate_bits, unsigned cap_height, const int32_t* arity_bits,
                           size_t n_layers, unsigned num_queries);
p25_status p25_fri_prove(const uint64_t* coeffs, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                         const int32_t* arity_bits, size_t n_layers, unsigned pow_bits, unsigned num_queries,
                         const uint64_t* seed, size_t n_seed, uint64_t* out, size_t out_cap, p25_status* status_out);

/* ------------------------------------------------------------------------------------------
 * Data formats either side of the path.
 * ------------------------------------------------------------------------------------------ */
/* plonky3 proof JSON (serde form of src/p3/serde/proof.rs:349-355, e.g. artifacts/proof_fibonacci.json)
 * -> input vector + shape.  Replaces serde_json::from_str::<P3ProofField> + set_witness
 * (src/p3/mod.rs:233-234, 254-257).  inputs_out may be NULL to query *n_out. */
p25_status p25_p3_proof_from_json(const char* json, size_t len, uint64_t* inputs_out, size_t cap, size_t* n_out,
                                  p25_p3_config* cfg_out);
/* Native plonky3 prover for the Fibonacci AIR of src/p3/mod.rs:160-221 (host code):
 * produces the hot path's per-proof input for any trace height 2^log_n (the reference ships exactly
 * one such proof, artifacts/proof_fibonacci.json = log_n 6, 100 queries, 16 PoW bits, which this
 * prover reproduces bit for bit).  pow_start: first proof-of-work witness tried (any valid witness is
 * a legitimate proof; different witnesses give different query indices).  inputs_out may be NULL to
 * query *n_out.  The shape (p25_p3_config) for p25_circuit_build_p3_verifier is returned in cfg_out. */
p25_status p25_p3_prove_fibonacci(int32_t log_n, int32_t num_queries, int32_t pow_bits, uint64_t pow_start,
                                  int32_t threads, uint64_t* inputs_out, size_t cap, size_t* n_out,
                                  p25_p3_config* cfg_out);
/* Input vector -> plonky3 proof JSON in the reference's serde format (src/p3/serde/proof.rs:16-355). */
p25_status p25_p3_inputs_to_json(const uint64_t* inputs, size_t n, const p25_p3_config* cfg, char* buf,
                                 size_t cap, size_t* len_out);
/* Flat proof -> JSON shaped like serde_json::to_string(&ProofWithPublicInputs) (src/p3/mod.rs:261).
 * buf may be NULL to query *len_out. */
p25_status p25_proof_to_json(p25_circuit* c, const uint64_t* proof, char* buf, size_t cap, size_t* len_out);

/* Flat proof <-> upstream's binary form `ProofWithPublicInputs::to_bytes()` / `from_bytes()` (plonky2 util/serialization.rs
 * write_proof_with_public_inputs, restated -- the crate is absent, so unpinned like the JSON): every field element as 8
 * little-endian bytes in the order of the flat layout, each Merkle proof preceded by its sibling count as one byte,
 * no public inputs.  buf may be NULL to query *len_out.  from_bytes rejects truncated / trailing / non-canonical data. */
p25_status p25_proof_to_bytes(p25_circuit* c, const uint64_t* proof, uint8_t* buf, size_t cap, size_t* len_out);
p25_status p25_proof_from_bytes(p25_circuit* c, const uint8_t* bytes, size_t len, uint64_t* proof_out, size_t cap_words);

#ifdef __cplusplus
}
#endif
#endif /* P25_H */
