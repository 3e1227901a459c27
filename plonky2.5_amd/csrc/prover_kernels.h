// Internal: device-side data structures and kernel launchers of the per-proof pipeline
// (witness -> wires commit -> Z/partial products -> quotient -> openings -> FRI).
#pragma once
#include "kernels.h"
#include "witness_program.h"

namespace p25 {

struct DeviceWitnessProgram {
  WitGen* d_gens = nullptr;
  uint32_t* d_args = nullptr;
  uint32_t* d_input_slots = nullptr;
  uint32_t* d_input_first = nullptr;
  uint32_t* d_wire_slot_cm = nullptr;
  std::vector<uint32_t> level_start;
  // the permutation generators of each level that run cooperatively for small batches (contiguous: gens are sorted
  // by kind within a level): Poseidon2Gate's (inner circuits) or, where a level has none, PoseidonGate's (recursion)
  std::vector<uint32_t> level_p2_begin, level_p2_count, level_coop_kind;
  std::vector<uint32_t> level_perm_arg_base;  // WitGen::arg_off of the level's first permutation generator (fixed stride after it)
  uint32_t* d_pi_slots = nullptr;
  uint32_t n_public_inputs = 0;
  uint32_t n_inputs = 0, num_slots = 0, num_random_fill = 0;
  size_t n_wire_elems = 0;
};
// Per-proof status word: the FIRST failure (in stream order) is the one reported -- a later kernel never
// overwrites an earlier code (upstream stops at its first panic / Err too).
#ifdef __HIPCC__
__device__ __forceinline__ void set_status(uint32_t* status, uint32_t code) { atomicCAS(status, 0u, code); }
#endif
// d_filler (nullable): explicit RandomValueGenerator values [n_proofs][num_random_fill] used instead of the seeds
// Proof p of the pass reads its inputs at d_inputs + min((p0 + p) * in_stride, in_max_off)  (in_stride 0 = n_inputs).
void launch_witgen(const DeviceWitnessProgram& wp, const u64* d_inputs, const u64* d_seeds, u64* d_vals,
                   size_t B, uint32_t n_proofs, uint32_t* d_status, hipStream_t st, const u64* d_filler = nullptr,
                   size_t in_stride = 0, size_t in_max_off = (size_t)-1, size_t p0 = 0);
void launch_fill_wires(const DeviceWitnessProgram& wp, const u64* d_vals, size_t B, uint32_t p, u64* d_wires,
                       hipStream_t st);

// Fiat-Shamir transcript kept on the device (upstream iop/challenger.rs `Challenger`): one wave,
// lanes 0..11 hold the sponge state; the permutation is computed cooperatively across lanes.
struct Transcript {
  u64 state[12];
  u64 in[8];
  u64 out[8];
  uint32_t n_in, n_out;
};
// init != 0: reset the transcript first.  Observes obs[0..n_obs), then draws n_chal challenges.
void launch_transcript(Transcript* d_tr, int init, const u64* d_obs, uint32_t n_obs, u64* d_chal_out,
                       uint32_t n_chal, hipStream_t st);
// observe two buffers back to back (e.g. openings at zeta, then at g*zeta)
void launch_pow_search(const Transcript* d_tr, int pow_bits, u64* d_result, hipStream_t st);
// Public inputs of proof p of a witness pass: values[i] = vals[pi_slots[i] * B + p] -> d_values_out[n] (the flat
// proof's public_inputs section) and their hash_no_pad (upstream `C::InnerHasher::hash_no_pad(&public_inputs)`) ->
// d_hash_out[4].  One wave: the sponge is a chain of ceil(n / 8) permutations.
void launch_public_inputs(const u64* d_vals, size_t B, uint32_t p, const uint32_t* d_pi_slots, uint32_t n, u64* d_values_out,
                          u64* d_hash_out, hipStream_t st);

// challenge block layout (u64 words) in the per-proof device scratch
enum {
  CH_BETAS = 0, CH_GAMMAS = 2, CH_ALPHAS = 4, CH_ZETA = 6, CH_FRI_ALPHA = 8, CH_FRI_BETAS = 10 /* 2 per layer, <= 8 layers */,
  CH_POW_WITNESS = 26, CH_POW_RESPONSE = 27, CH_QUERIES = 28 /* <= 64 */, CH_WORDS = 96
};

struct GateEntry {
  uint32_t kind, selector_index, group_start, group_end;
};
struct QuotientArgs {
  const u64* cs_lde;     // [num_cs][big]  selectors | constants | sigmas, bit-reversed positions
  const u64* wires_lde;  // [num_wires][big]
  const u64* zs_lde;     // [NC*(1+NP)][big]
  const u64* pow_big;    // w_big^e, e < big
  const u64* k_is;       // [num_routed]
  const u64* chal;       // challenge block
  const u64* alpha_pows; // [2][ALPHA_POWS]
  u64* out;              // [NC][big] quotient values at bit-reversed positions
  GateEntry gates[16];
  uint32_t n_gates, num_selectors, num_wires, num_routed, num_partial_products, degree_bits, rate_bits;
  uint32_t quotient_degree_factor;  // routed wires per partial-product chunk (max_quotient_degree_factor)
  u64 zh[8], zh_inv[8];  // Z_H on the coset (index = i mod 2^rate_bits), and inverses
  const u64* l0_inv;     // [big] 1 / (n (x - 1)) at bit-reversed positions (per circuit)
  const u64* pi_hash;    // [4] public-inputs hash of this proof (PublicInputGate: wire_i - hash_i)
};
void launch_l0_inv(const u64* d_pow_big, uint32_t degree_bits, uint32_t rate_bits, u64* d_out, hipStream_t st);
void launch_alpha_pows(const u64* d_chal, u64* d_alpha_pows, hipStream_t st);
void launch_quotient(const QuotientArgs& a, hipStream_t st);

// Z and partial products (values in natural row order): out[NC*(1+NP)][n]
struct ZppArgs {
  const u64* wires;      // [num_wires][n] witness values
  const u64* sigmas;     // [num_routed][n] sigma values
  const u64* pow_n;      // w_n^r
  const u64* k_is;
  const u64* chal;
  u64* chunk;            // scratch [NC][NP+1][n]
  u64* tot;              // scratch [NC][n]
  u64* block_tot;        // scratch [NC][n/256]
  u64* out;
  uint32_t n, num_routed, num_partial_products, num_challenges, quotient_degree_factor;
};
void launch_zpp(const ZppArgs& a, hipStream_t st);

// openings: evaluates n_polys coefficient vectors (length n, stride n) at the extension point read
// from d_point[0..2) (optionally multiplied by `scale`), writing (a, b) pairs to out[2*n_polys].
void launch_eval_polys(const u64* d_coeffs, uint32_t n_polys, uint32_t log_n, const u64* d_point, u64 scale,
                       u64* d_scratch_pows /*[2*1026 + 2*n_polys*chunks]*/, u64* d_out, hipStream_t st, bool reuse_pows = false);

// FRI
struct FriCombineArgs {
  const u64* coeffs[4];   // the 4 oracles' coefficient matrices [n_polys][n]
  uint32_t n_polys[4];
  uint32_t log_n, num_challenges;
  const u64* chal;        // CH_ZETA, CH_FRI_ALPHA
  u64 g;                  // generator of the size-n subgroup (zeta_next = g*zeta)
  u64* comp;              // scratch [2 batches][2 comps][n]
  u64* scan_tmp;          // scratch [2][2][n] + block totals
  u64* final_a;           // out: final polynomial coefficients, component a [n]
  u64* final_b;           //      component b [n]
};
void launch_fri_combine(const FriCombineArgs& a, hipStream_t st);
// folded[i] = sum_k beta^k c[arity*i + k]; component arrays
void launch_fri_fold(const u64* ca, const u64* cb, uint32_t len_out, uint32_t arity_bits, const u64* d_beta,
                     u64* oa, u64* ob, hipStream_t st);
// leaves: 2^arity_bits consecutive extension values (bit-reversed order), flattened (a, b)
void launch_fri_leaf_hash(const u64* va, const u64* vb, uint32_t n_leaves, uint32_t arity_bits, u64* d_digests,
                          hipStream_t st, bool single_proof = false);
void launch_tree_from_digests(u64* d_tree, size_t n_leaves, unsigned cap_height, hipStream_t st,
                              bool single_proof = false);

// query phase: writes initial-tree openings and FRI steps into the flat proof
struct QueryArgs {
  const u64* chal;            // CH_QUERIES raw challenges
  uint32_t num_queries, lde_bits, cap_height;
  uint32_t n_oracles;         // initial oracles opened per query (4 in a plonky2 proof, 0 for FRI alone)
  const u64* oracle_lde[4];   // [width][big]
  const u64* oracle_tree[4];
  uint32_t oracle_width[4];
  uint32_t n_layers;
  uint32_t arity_bits[8];
  const u64* layer_va[8];
  const u64* layer_vb[8];
  const u64* layer_tree[8];
  u64* proof;                 // flat proof buffer
  uint32_t query_offset, query_stride;
};
void launch_queries(const QueryArgs& a, hipStream_t st);

}  // namespace p25
