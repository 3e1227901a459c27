// Internal declarations shared by the HIP translation units of libp25 (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdexcept>
#include <string>
#include <vector>
#include <map>
#include <tuple>
#include "gl.h"

// Wave priority (s_setprio, 0..3) in the SIMDs' issue arbitration.  The bulk hash kernels (k_hash_leaves*, k_tree_level,
// k_pow_search: 68 % of all VALU work, always enough of it resident to fill every issue slot) stay at 0; the kernels whose
// waves spend their lives waiting for memory, LDS or barriers (NTT, quotient, partial products, openings, FRI) run at 2
// and the single-wave chains (transcript, cooperative Merkle tops, witness levels) at 3, so that when they CAN issue they
// do, finish, and give their registers and LDS back.  Measured: profiles/r03_pipeline_model_experiments.txt item 16.
#if defined(__HIP_DEVICE_COMPILE__)
#define P25_WAVE_PRIO(n) __builtin_amdgcn_s_setprio(n)
#else
#define P25_WAVE_PRIO(n) ((void)0)
#endif
#define P25_PRIO_BULK 2
#define P25_PRIO_CHAIN 3
namespace p25 {

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
#define P25_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw p25::HipError(std::string(#expr) + ": " + hipGetErrorString(_e) + " at " +        \
                          __FILE__ + ":" + std::to_string(__LINE__));                         \
  } while (0)

// ---------------- hashing (kernels_hash.hip) ----------------
void launch_poseidon_permute(u64* d_states, size_t n, hipStream_t st);
void launch_poseidon2_permute(u64* d_states, size_t n, hipStream_t st);
size_t merkle_tree_words(size_t n_leaves, unsigned cap_height);
size_t merkle_level_offset(size_t n_leaves, unsigned level);
// ev_begin/ev_end (optional) bracket the leaf-sponge kernel -- the dominant kernel of the prover --
// so bench.py can report its measured duration (roofline line).
// single_proof: only one proof is in flight (latency matters more than instruction count): cooperative
// kernels for larger levels.
u64* launch_merkle_tree(const u64* d_cols, size_t col_stride, int width, size_t n_leaves,
                        unsigned cap_height, u64* d_tree, hipStream_t st, hipEvent_t ev_begin = nullptr,
                        hipEvent_t ev_end = nullptr, bool single_proof = false);

// ---------------- NTT (kernels_ntt.hip) ----------------
// One pass of a two-pass (four-step) NTT: a block transforms a tile of 2^log_t independent
// length-2^log_r sub-transforms held in LDS.  See kernels_ntt.hip for the addressing kinds.
struct NttPass {
  const u64* in;
  u64* out;
  size_t in_poly_stride, out_poly_stride;
  size_t coset_out_off[8];
  int log_r, log_nt, log_t;
  int in_kind, in_br_t, in_br_i;
  int out_kind, out_br_t, out_br_i;
  const u64* pow_table;  // w_N^e (forward) or w_N^-e (inverse), e < N = 2^log_n_table
  int log_n_table;
  int use_twiddle;       // multiply output (t, j) by pow_table[t*j] (N = R*NT required)
  const u64* pre;        // optional [n_cosets][N]: input element at address k is multiplied by pre[coset][k]
  const u64* pre_t;      // or, for N too large for that table to stay in L2, its two factors (in_kind 0, natural order only):
  const u64* pre_i;      //   pre_t[coset][t] * pre_i[coset][i] for the element (t, i) at address t + i * NT
  const u64* post_t;     // optional [NT]
  const u64* post_i;     // optional [R]
  int inverse;           // pow_table holds powers of the INVERSE root (selects the 16th-root constants of kernels_ntt.hip)
  int lazy_out;          // the output is read by another pass of k_ntt_tile only: store any u64 congruent to the value
                         // (the kernel computes in lazy arithmetic, gl_lazy.h); 0 = canonical, as everything else expects
  uint32_t n_tiles, n_cosets, log_g, full_table;  // set by launch_ntt_pass (log_g: block decode of k_ntt_tile)
};
void launch_ntt_pass(const NttPass& p, int n_polys, int n_cosets, hipStream_t st);
// Shader clock (Hz) under a full-chip Poseidon load, from in-kernel cycle and wall-clock counters (kernels_hash.hip).
double measure_shader_clock_hz(hipStream_t st);

class NttTables {
 public:
  ~NttTables();
  // device table of w^e, e < 2^log_n, w = primitive 2^log_n-th root (or its inverse)
  const u64* pow_table(int log_n, bool inverse);
  // device table of base^e for e < len (cached by (base, len))
  const u64* geom_table(u64 first, u64 ratio, size_t len);
  const u64* upload(const std::vector<u64>& host);
  typedef std::map<std::tuple<int, int, u64>, const u64*> CosetCache;
  CosetCache& coset_cache() { return coset_; }

 private:
  CosetCache coset_;
  std::map<std::pair<int, bool>, u64*> pow_;
  std::map<std::tuple<u64, u64, size_t>, u64*> geom_;
  std::vector<u64*> owned_;
};

// values (natural order) -> coefficients (natural order); in/out/tmp are [n_polys][n] with the
// given strides; tmp must not alias in or out (in may alias tmp's role only if in == tmp is never
// reused).  If `coset_shift` != 1 computes the coset iNTT (coefficients of f given f(shift*w^i)).
void ntt_inverse(NttTables& tb, const u64* d_in, size_t in_stride, bool in_bitrev, u64* d_tmp,
                 size_t tmp_stride, u64* d_out, size_t out_stride, int log_n, int n_polys,
                 u64 coset_shift, hipStream_t st);
// coefficients (natural order, n each) -> values of the rate-2^rate_bits LDE on the coset
// shift*<w_{n*2^rate_bits}>, stored at BIT-REVERSED index (position rev(i) holds f(shift*w^i)):
// exactly the leaf order of upstream's PolynomialBatch (transpose + reverse_index_bits).
void ntt_lde_bitrev(NttTables& tb, const u64* d_coeffs, size_t coeff_stride, u64* d_lde,
                    size_t lde_stride, int log_n, int rate_bits, int n_polys, u64 shift,
                    hipStream_t st);

// values -> coefficients (kept) -> LDE: ntt_inverse followed by ntt_lde_bitrev.  d_tmp must not alias anything.
void ntt_inverse_then_lde(NttTables& tb, const u64* d_vals, size_t val_stride, u64* d_tmp, size_t tmp_stride, u64* d_coeffs,
                          size_t coeff_stride, u64* d_lde, size_t lde_stride, int log_n, int rate_bits, int n_polys, u64 shift,
                          hipStream_t st);

}  // namespace p25
