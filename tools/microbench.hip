// Integer-throughput microbenchmarks for gfx950 that size the Goldilocks/Poseidon kernels.
// Build: hipcc -O3 --offload-arch=gfx950 -I../plonky2.5_amd/csrc microbench.hip -o build/microbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "poseidon.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_op(u64* out, int iters, u32 seed) {
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  u64 a0 = tid * 0x9E3779B97F4A7C15ull + seed, a1 = a0 ^ 0x1234567ull, a2 = a0 + 77, a3 = a0 * 3 + 1;
  u32 m = (u32)(a0 >> 13) | 1;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (OP == 0) {  // mad_u64_u32: 32x32+64
        a0 = (u64)(u32)a0 * m + a1; a1 = (u64)(u32)a1 * m + a2; a2 = (u64)(u32)a2 * m + a3; a3 = (u64)(u32)a3 * m + a0;
      } else if (OP == 1) {  // mul_lo_u32
        u32 x0 = (u32)a0 * m, x1 = (u32)a1 * m, x2 = (u32)a2 * m, x3 = (u32)a3 * m;
        a0 = x0 + 1; a1 = x1 + 2; a2 = x2 + 3; a3 = x3 + 5;
      } else if (OP == 2) {  // mul_hi_u32
        a0 = __umulhi((u32)a0, m) + 1u; a1 = __umulhi((u32)a1, m) + 3u; a2 = __umulhi((u32)a2, m) + 5u; a3 = __umulhi((u32)a3, m) + 7u;
      } else if (OP == 3) {  // 64-bit add (2 instr)
        a0 += a1; a1 += a2; a2 += a3; a3 += a0;
      } else if (OP == 4) {  // goldilocks mul_nc
        a0 = gl::mul_nc(a0, a1); a1 = gl::mul_nc(a1, a2); a2 = gl::mul_nc(a2, a3); a3 = gl::mul_nc(a3, a0);
      } else if (OP == 5) {  // mul24
        u32 x0 = __umul24((u32)a0, m), x1 = __umul24((u32)a1, m), x2 = __umul24((u32)a2, m), x3 = __umul24((u32)a3, m);
        a0 = x0 + 1; a1 = x1 + 2; a2 = x2 + 3; a3 = x3 + 5;
      } else if (OP == 6) {  // fp64 fma
        double d0 = __longlong_as_double(a0 | 0x3ff0000000000000ull), d1 = 1.000001, d2 = 0.5;
        double e0 = d0, e1 = d0 + 1, e2 = d0 + 2, e3 = d0 + 3;
        e0 = fma(e0, d1, d2); e1 = fma(e1, d1, d2); e2 = fma(e2, d1, d2); e3 = fma(e3, d1, d2);
        a0 = __double_as_longlong(e0); a1 = __double_as_longlong(e1); a2 = __double_as_longlong(e2); a3 = __double_as_longlong(e3);
      } else if (OP == 7) {  // canonical goldilocks mul
        a0 = gl::mul(a0, a1); a1 = gl::mul(a1, a2); a2 = gl::mul(a2, a3); a3 = gl::mul(a3, a0);
      }
    }
  }
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}

__global__ __launch_bounds__(256) void k_perm(u64* states, size_t n, int reps) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
  for (int k = 0; k < 12; k++) s[k] = states[k * n + i];
  for (int r = 0; r < reps; r++) poseidon::permute(s);
  for (int k = 0; k < 12; k++) states[k * n + i] = s[k];
}

template <int OP>
int run(const char* name, double ops_per_iter) {
  const int blocks = 256 * 16, threads = 256, iters = 2000;
  u64* d;
  CK(hipMalloc(&d, (size_t)blocks * threads * 8));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, d, 10, 1u);
  CK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, d, iters, 2u);
  hipEventRecord(e1);
  CK(hipDeviceSynchronize());
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double total = (double)blocks * threads * iters * 8 * ops_per_iter;
  printf("%-22s %8.3f ms  %10.2f Gop/s  (%.3f ops/clk/CU @2.4GHz)\n", name, ms, total / ms / 1e6,
         total / (ms * 1e-3) / 256 / 2.4e9);
  hipFree(d);
  return 0;
}

int main() {
  int n;
  CK(hipGetDeviceCount(&n));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, CUs %d, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  run<0>("mad_u64_u32", 4);
  run<1>("mul_lo_u32", 4);
  run<2>("mul_hi_u32", 4);
  run<3>("add_u64", 4);
  run<5>("mul_u32_u24", 4);
  run<6>("fma_f64", 4);
  run<4>("goldilocks mul_nc", 4);
  run<7>("goldilocks mul canon", 4);
  {
    size_t np = (size_t)1 << 22;
    u64* d;
    CK(hipMalloc(&d, np * 96));
    CK(hipMemset(d, 1, np * 96));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_perm, dim3(np / 256), dim3(256), 0, 0, d, np, 1);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_perm, dim3(np / 256), dim3(256), 0, 0, d, np, 4);
    hipEventRecord(e1);
    CK(hipDeviceSynchronize());
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("poseidon permute: %.3f ms for %zu perms -> %.1f Mperm/s\n", ms, np * 4, np * 4 / ms / 1e3);
  }
  return 0;
}
