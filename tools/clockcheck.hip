// Shader clock under the Poseidon load: s_memtime (shader cycles) against the constant-rate wall clock, per block,
// for the VALU permutation (poseidon.h) and for the wave-wide one with the MDS layers on the matrix cores
// (poseidon_mfma.h): cycles per permutation AND the clock the chip sustains while running it.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "poseidon.h"
#include "poseidon_mfma.h"
template <int MX, int MINW>
__global__ __launch_bounds__(64, MINW) void k(u64* out, int reps, unsigned long long* clk) {
#if defined(__HIP_DEVICE_COMPILE__)
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = l * 12 + i;
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(threadIdx.x);
  unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int r = 0; r < reps; r++) {
    s[0] ^= r;
    if (MX) poseidon::mx::permute_wave(s, poseidon::ROWS_ALL, c); else poseidon::permute(s);
  }
  unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  for (int i = 0; i < 4; i++) out[4 * l + i] = s[i];
  if (threadIdx.x == 0 && blockIdx.x % 1024 == 0) { clk[2 * (blockIdx.x / 1024)] = c1 - c0; clk[2 * (blockIdx.x / 1024) + 1] = w1 - w0; }
#endif
}
int main() {
  const size_t n = 1 << 19; u64* d; unsigned long long* c;
  hipMalloc(&d, n * 32); hipMallocManaged(&c, 64 * 16);
  int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
  printf("wall clock rate %d kHz\n", rate);
  const int reps = 17 * 4;
  for (int var = 0; var < 4; var++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; it++) {
      hipEventRecord(e0);
      if (var == 0) hipLaunchKernelGGL((k<0, 6>), dim3(n / 64), dim3(64), 0, 0, d, reps, c);
      if (var == 1) hipLaunchKernelGGL((k<0, 4>), dim3(n / 64), dim3(64), 0, 0, d, reps, c);
      if (var == 2) hipLaunchKernelGGL((k<1, 4>), dim3(n / 64), dim3(64), 0, 0, d, reps, c);
      if (var == 3) hipLaunchKernelGGL((k<1, 3>), dim3(n / 64), dim3(64), 0, 0, d, reps, c);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    double cyc = 0, wall = 0;
    for (int i = 0; i < 8; i++) { cyc += c[2 * i]; wall += c[2 * i + 1]; }
    const char* names[4] = {"VALU permutation (64,6)", "VALU permutation (64,4)", "MFMA permutation (64,4)", "MFMA permutation (64,3)"};
    printf("%-26s %8.3f ms  %7.1f Mperm/s   %8.0f shader cycles per permutation per wave (wave lifetime)   clock %.3f GHz\n", names[var], ms,
           (double)n * reps / ms / 1e3, cyc / 8 / reps, cyc / (wall / (rate * 1e3)) / 1e9);
  }
  return 0;
}
