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
  P25_ERR_PARSE = 8
};

/* Last error message of the calling thread ("" if none). */
const char* p25_last_error(void);
/* Library version string. */
const char* p25_version(void);
/* Select the HIP device used by this process (one process per GPU).  P25_ERR_NO_DEVICE if none. */
p25_status p25_device_init(int device_index);

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

#ifdef __cplusplus
}
#endif
#endif /* P25_H */
