// Permutation argument: partial products and Z (upstream plonky2 @ 3de92d9 prover.rs
// `wires_permutation_partial_products_and_zs`, plonk_common.rs `partial_products_and_z_gx`;
// SURVEY.md App. A.5; reached from /root/reference/src/p3/mod.rs:260).
//
// For challenge c and row r (x = g^r):   q_j = (w_j + beta*k_j*x + gamma) / (w_j + beta*sigma_j + gamma)
// chunk_k = prod of 8 consecutive q_j; Z(g x) = Z(x) * prod_k chunk_k; pp_k(x) = Z(x) * prod_{j<=k} chunk_j.
// Output polynomial order: [Z_0, Z_1, pp_0[0..NP), pp_1[0..NP)].
//
// Kernel 1 (one lane per (row, challenge)): numerator / denominator products per chunk, one batched
//   inversion of the 10 chunk denominators (Montgomery trick) instead of 80 inversions -- the field
//   value N_k / D_k equals upstream's product of per-wire quotients exactly.
// Kernels 2-4: running product over rows as a three-phase scan (block scan in LDS, scan of block
//   totals, apply), replacing upstream's sequential loop over 2^16 rows.
#include "kernels.h"
#include "gl_lazy.h"
#include "prover_kernels.h"

namespace p25 {


__global__ __launch_bounds__(256) void k_zpp_chunks(ZppArgs a) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t c = blockIdx.y;
  if (r >= a.n) return;
  const u64 beta = a.chal[CH_BETAS + c], gamma = a.chal[CH_GAMMAS + c];
  const u64 x = a.pow_n[r];
  const int nch = (int)a.num_partial_products + 1;
  const int per = (int)a.quotient_degree_factor;  // routed wires per chunk (max_quotient_degree_factor)
  u64 num[MAX_CHUNKS], den[MAX_CHUNKS];
  for (int k = 0; k < nch; k++) {
    u64 np = 1, dp = 1;
    for (int j = k * per; j < (k + 1) * per && j < (int)a.num_routed; j++) {
      // lazy arithmetic (gl_lazy.h): w + gamma with one correction (both canonical), the two affine forms as fused
      // multiply-adds, the running products on any-u64 values; canonical once per chunk
      const u64 wg = gl::add_c(a.wires[(size_t)j * a.n + r], gamma);
      const u64 nu = gl::mad_nc(beta, gl::mul_nc(a.k_is[j], x), wg);
      const u64 de = gl::mad_nc(beta, a.sigmas[(size_t)j * a.n + r], wg);
      np = gl::mul_nc(np, nu);
      dp = gl::mul_nc(dp, de);
    }
    num[k] = gl::canon(np);
    den[k] = gl::canon(dp);
  }
  // batch inverse of den[0..nch)
  u64 pre[MAX_CHUNKS];
  u64 acc = 1;
  for (int k = 0; k < nch; k++) {
    pre[k] = acc;
    acc = gl::mul(acc, den[k]);
  }
  u64 inv = gl::inv(acc);
  u64 tot = 1;
  u64 q[MAX_CHUNKS];
  for (int k = nch - 1; k >= 0; k--) {
    u64 di = gl::mul(inv, pre[k]);
    inv = gl::mul(inv, den[k]);
    q[k] = gl::mul(num[k], di);
  }
  for (int k = 0; k < nch; k++) {
    a.chunk[((size_t)c * nch + k) * a.n + r] = q[k];
    tot = gl::mul(tot, q[k]);
  }
  a.tot[(size_t)c * a.n + r] = tot;
}

// inclusive product scan within 256-row blocks; block totals to block_tot
__global__ __launch_bounds__(256) void k_zpp_scan_block(ZppArgs a) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 sh[256];
  uint32_t c = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  u64 v = r < a.n ? a.tot[(size_t)c * a.n + r] : 1;
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    u64 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 1;
    __syncthreads();
    v = gl::mul(v, t);
    sh[threadIdx.x] = v;
    __syncthreads();
  }
  if (r < a.n) a.tot[(size_t)c * a.n + r] = v;
  if (threadIdx.x == 255) a.block_tot[(size_t)c * gridDim.x + blockIdx.x] = v;
}
// exclusive product scan of the block totals (sequential over 256-wide tiles; one block per challenge)
__global__ __launch_bounds__(256) void k_zpp_scan_totals(u64* block_tot, uint32_t n_blocks) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 sh[256];
  __shared__ u64 carry_s;
  u64* bt = block_tot + (size_t)blockIdx.x * n_blocks;
  if (threadIdx.x == 0) carry_s = 1;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 256) {
    uint32_t i = base + threadIdx.x;
    u64 orig = i < n_blocks ? bt[i] : 1;
    u64 v = orig;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      u64 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 1;
      __syncthreads();
      v = gl::mul(v, t);
      sh[threadIdx.x] = v;
      __syncthreads();
    }
    u64 carry = carry_s;
    // exclusive = carry * inclusive / orig  -> use the neighbour's inclusive value instead of dividing
    u64 excl = threadIdx.x == 0 ? carry : gl::mul(carry, sh[threadIdx.x - 1]);
    __syncthreads();
    if (i < n_blocks) bt[i] = excl;
    if (threadIdx.x == 255) carry_s = gl::mul(carry, v);
    __syncthreads();
  }
}
// Z(r) = (product of all rows before r); pp_k(r) = Z(r) * prod_{j<=k} chunk_j(r)
__global__ __launch_bounds__(256) void k_zpp_finish(ZppArgs a) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  uint32_t c = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  if (r >= a.n) return;
  const int NP = (int)a.num_partial_products, nch = NP + 1;
  u64 z = a.block_tot[(size_t)c * gridDim.x + blockIdx.x];
  if (threadIdx.x > 0) z = gl::mul(z, a.tot[(size_t)c * a.n + r - 1]);
  a.out[(size_t)c * a.n + r] = z;
  u64 acc = z;
  for (int k = 0; k < NP; k++) {
    acc = gl::mul(acc, a.chunk[((size_t)c * nch + k) * a.n + r]);
    a.out[((size_t)a.num_challenges + (size_t)c * NP + k) * a.n + r] = acc;
  }
}

void launch_zpp(const ZppArgs& a, hipStream_t st) {
  dim3 grid((a.n + 255) / 256, a.num_challenges);
  hipLaunchKernelGGL(k_zpp_chunks, grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_zpp_scan_block, grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_zpp_scan_totals, dim3(a.num_challenges), dim3(256), 0, st, a.block_tot, grid.x);
  hipLaunchKernelGGL(k_zpp_finish, grid, dim3(256), 0, st, a);
}

}  // namespace p25
