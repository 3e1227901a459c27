// Shader clock under the Poseidon load: s_memtime (shader cycles) against the 100 MHz wall clock, per block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "poseidon.h"
__global__ __launch_bounds__(64) void k(u64* out, int reps, unsigned long long* clk) {
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = l * 12 + i;
  unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int r = 0; r < reps; r++) { s[0] ^= r; poseidon::permute(s); }
  unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  for (int i = 0; i < 4; i++) out[4 * l + i] = s[i];
  if (threadIdx.x == 0 && blockIdx.x % 1024 == 0) { clk[2 * (blockIdx.x / 1024)] = c1 - c0; clk[2 * (blockIdx.x / 1024) + 1] = w1 - w0; }
}
int main() {
  const size_t n = 1 << 19; u64* d; unsigned long long* c;
  hipMalloc(&d, n * 32); hipMallocManaged(&c, 64 * 16);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  for (int it = 0; it < 3; it++) {
    hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, d, 17 * 4, c);
    hipDeviceSynchronize();
  }
  printf("wall clock rate %d kHz\n", rate);
  for (int i = 0; i < 8; i++) printf("block %d: shader cycles %llu, wall ticks %llu -> %.3f GHz\n", i * 1024, c[2 * i], c[2 * i + 1], (double)c[2 * i] / ((double)c[2 * i + 1] / (rate * 1e3)) / 1e9);
  return 0;
}
