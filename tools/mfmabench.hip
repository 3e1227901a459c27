// What does a v_mfma_i32_32x32x32_i8 cost the VALU stream around it?  Loops of one MFMA (rotating over NACC
// independent accumulators) followed by K independent v_mad_u64_u32, at 1..4 waves per SIMD; cycles per loop
// iteration per SIMD from the wall clock and the measured shader clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
typedef unsigned int u32;

template <int K, int MF, int CHAIN>
__global__ __launch_bounds__(256) void k_loop(int* out, int iters, long long* clk) {
#if defined(__HIP_DEVICE_COMPILE__)
  v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
  v16i acc[4];
  for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) acc[j][r] = r + j;
  u64 m[16];
  for (int j = 0; j < 16; j++) m[j] = threadIdx.x + j;
  u32 x = threadIdx.x | 1, y = 12345;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (MF) {
        if (CHAIN) acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[0], 0, 0, 0);
        else acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < K; k++) {
        u64 dm;
        asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(m[k % 16]), "=s"(dm) : "v"(x), "v"(y));
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int j = 0; j < 4; j++) for (int r = 0; r < 16; r++) s += acc[j][r];
  u64 ms = 0;
  for (int j = 0; j < 16; j++) ms += m[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)ms;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
#endif
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int* g_out; long long* g_clk;
template <int K, int MF, int CHAIN> void run(int waves_per_simd) {
  const int iters = 20000;
  // one block of 256 lanes = one wave per SIMD of a CU; `waves_per_simd` blocks per CU, 256 CUs
  int blocks = 256 * waves_per_simd;
  float ms = timeit([&] { hipLaunchKernelGGL((k_loop<K, MF, CHAIN>), dim3(blocks), dim3(256), 0, 0, g_out, iters, g_clk); });
  long long c; (void)hipMemcpy(&c, g_clk, 8, hipMemcpyDeviceToHost);
  // s_memtime / readcyclecounter ticks at 100 MHz on gfx9 (constant rate): convert wall-clock via ms instead
  double per_iter_ns = ms * 1e6 / ((double)iters * 4);          // per (MFMA + K mads) group, per wave
  double per_simd_ns = per_iter_ns / 1.0;                        // waves on a SIMD run concurrently: time per group per wave
  printf("K=%2d mfma=%d chain=%d waves/SIMD=%d : %8.3f ms  %7.2f ns per group per wave  -> %7.2f ns of SIMD time per group\n", K, MF, CHAIN,
         waves_per_simd, ms, per_iter_ns, per_simd_ns / waves_per_simd);
}

int main() {
  CK(hipMalloc(&g_out, 256 * 8 * 256 * 4)); CK(hipMalloc(&g_clk, 8));
  for (int w : {1, 2, 4}) {
    run<0, 1, 0>(w); run<0, 1, 1>(w);
    run<4, 0, 0>(w); run<8, 0, 0>(w); run<16, 0, 0>(w);
    run<2, 1, 0>(w); run<4, 1, 0>(w); run<6, 1, 0>(w); run<8, 1, 0>(w); run<12, 1, 0>(w); run<16, 1, 0>(w); run<24, 1, 0>(w);
    run<8, 1, 1>(w); run<16, 1, 1>(w);
  }
  return 0;
}
