// Latency of one chained cooperative Poseidon permutation on a single wave: the shuffle form (coop.h),
// a variant with two accumulator pairs, and the single-state form that broadcasts through SGPRs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "coop.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
// V0: current cooperative permutation, chained
__global__ __launch_bounds__(64) void k_v0(u64* io, int reps) {
  __shared__ u64 rc[360];
  coop::stage_poseidon_rc(rc);
  int lane = threadIdx.x;
  u64 s = lane < 12 ? io[lane] : 0;
  for (int r = 0; r < reps; r++) s = coop::poseidon_permute(s, lane, rc);
  if (lane < 12) io[lane] = s;
}
// V3: single group per wave; state words broadcast through SGPRs (v_readlane), coefficients per lane
__device__ inline u64 perm_v3(u64 s, int lane, const u64* __restrict__ rc) {
  const int r = lane < 12 ? lane : 0;
  u32 coef[12];
#pragma unroll
  for (int j = 0; j < 12; j++) coef[j] = poseidon::MDS_CIRC[(j - r + 12) % 12] + ((r == 0 && j == 0) ? poseidon::MDS_DIAG0 : 0);
  for (int rd = 0; rd < poseidon::N_ROUNDS; rd++) {
    u64 t = poseidon::add_rc(s, rc[12 * rd + r]);
    bool full = rd < poseidon::HALF_FULL || rd >= poseidon::HALF_FULL + poseidon::N_PARTIAL;
    u64 sb = poseidon::sbox(t);
    s = (full || r == 0) ? sb : t;
    u32 lo = (u32)s, hi = (u32)(s >> 32);
    u64 al0 = 0, ah0 = 0, al1 = 0, ah1 = 0;
#pragma unroll
    for (int j = 0; j < 12; j += 2) {
      u32 l0 = __builtin_amdgcn_readlane(lo, j), h0 = __builtin_amdgcn_readlane(hi, j);
      u32 l1 = __builtin_amdgcn_readlane(lo, j + 1), h1 = __builtin_amdgcn_readlane(hi, j + 1);
      al0 += (u64)l0 * coef[j];
      ah0 += (u64)h0 * coef[j];
      al1 += (u64)l1 * coef[j + 1];
      ah1 += (u64)h1 * coef[j + 1];
    }
    u64 al = al0 + al1, ah = ah0 + ah1;
    u64 l64 = al + (ah << 32);
    u32 h32 = (u32)(ah >> 32) + (l64 < al ? 1u : 0u);
    s = gl::reduce96(l64, h32);
  }
  return gl::canon(s);
}
__global__ __launch_bounds__(64) void k_v3(u64* io, int reps) {
  __shared__ u64 rc[360];
  coop::stage_poseidon_rc(rc);
  int lane = threadIdx.x;
  u64 s = lane < 12 ? io[lane] : 0;
  for (int r = 0; r < reps; r++) s = perm_v3(s, lane, rc);
  if (lane < 12) io[lane] = s;
}
// V1: like V0 but two independent accumulator pairs
__device__ inline u64 perm_v1(u64 s, int lane, const u64* __restrict__ rc) {
  const int base = lane & ~15, rr = lane & 15;
  const int r = rr < 12 ? rr : 0;
  for (int rd = 0; rd < poseidon::N_ROUNDS; rd++) {
    u64 t = poseidon::add_rc(s, rc[12 * rd + r]);
    bool full = rd < poseidon::HALF_FULL || rd >= poseidon::HALF_FULL + poseidon::N_PARTIAL;
    u64 sb = poseidon::sbox(t);
    s = (full || r == 0) ? sb : t;
    u32 lo = (u32)s, hi = (u32)(s >> 32);
    u64 al0 = 0, ah0 = 0, al1 = 0, ah1 = 0;
#pragma unroll
    for (int i = 0; i < 12; i += 2) {
      int s0 = i + r, s1 = i + 1 + r;
      s0 = base + (s0 >= 12 ? s0 - 12 : s0);
      s1 = base + (s1 >= 12 ? s1 - 12 : s1);
      al0 += (u64)__shfl(lo, s0) * poseidon::MDS_CIRC[i];
      ah0 += (u64)__shfl(hi, s0) * poseidon::MDS_CIRC[i];
      al1 += (u64)__shfl(lo, s1) * poseidon::MDS_CIRC[i + 1];
      ah1 += (u64)__shfl(hi, s1) * poseidon::MDS_CIRC[i + 1];
    }
    if (r == 0) { al0 += (u64)lo * poseidon::MDS_DIAG0; ah0 += (u64)hi * poseidon::MDS_DIAG0; }
    u64 al = al0 + al1, ah = ah0 + ah1;
    u64 l64 = al + (ah << 32);
    u32 h32 = (u32)(ah >> 32) + (l64 < al ? 1u : 0u);
    s = gl::reduce96(l64, h32);
  }
  return gl::canon(s);
}
__global__ __launch_bounds__(64) void k_v1(u64* io, int reps) {
  __shared__ u64 rc[360];
  coop::stage_poseidon_rc(rc);
  int lane = threadIdx.x;
  u64 s = (lane & 15) < 12 ? io[lane & 15] : 0;
  for (int r = 0; r < reps; r++) s = perm_v1(s, lane, rc);
  if (lane < 12) io[lane] = s;
}
// the product's single-state form (coop.h)
__global__ __launch_bounds__(64) void k_single(u64* io, int reps) {
  __shared__ u64 rc[360];
  coop::stage_poseidon_rc(rc);
  int lane = threadIdx.x;
  u64 s = lane < 12 ? io[lane] : 0;
  for (int r = 0; r < reps; r++) s = coop::poseidon_permute_single(s, lane, rc);
  if (lane < 12) io[lane] = s;
}
template <class K> void run(const char* name, K k) {
  u64* d; (void)hipMallocManaged(&d, 128);
  for (int i = 0; i < 12; i++) d[i] = i;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 10); (void)hipDeviceSynchronize();
  for (int i = 0; i < 12; i++) d[i] = i;
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1001);
  (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %8.2f us per permutation   out[0]=%016llx\n", name, ms * 1e3 / 1001, (unsigned long long)d[0]);
}
int main() { run("v0 coop (shfl)", k_v0); run("v1 2 accumulator pairs", k_v1); run("v3 readlane/SGPR broadcast", k_v3); run("coop.h single-state form", k_single); return 0; }
