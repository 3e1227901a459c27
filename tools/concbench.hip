// How many kernels does the MI355X keep ACTIVE at once?  K streams, each M launches of a one-wave kernel that spins for T us
// of the constant-rate wall clock; wall time of the whole lot -> concurrency = K * M * T / wall.  (Round 5: the kernel trace of
// the 16-stream proving pipeline never shows more than 5-7 kernels executing together; is that the hardware?)
// usage: concbench [T_us=200] [M=20]      (GPU_MAX_HW_QUEUES from the environment, as the library sets it: 24)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_spin(unsigned long long ticks, unsigned long long* sink) {
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0x123456789ull) sink[0] = t0;
}
// a second flavour: `blocks` workgroups of 64 lanes (a "bulk" kernel that fills part of the chip), each spinning T us
int main(int argc, char** argv) {
  const double T_us = argc > 1 ? atof(argv[1]) : 200.0;
  const int M = argc > 2 ? atoi(argv[2]) : 20;
  int khz = 0;
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
  const unsigned long long ticks = (unsigned long long)(T_us * 1e-6 * khz * 1e3);
  unsigned long long* sink;
  CK(hipMalloc(&sink, 64));
  printf("wall clock %d kHz, spin %.0f us = %llu ticks, %d launches per stream, GPU_MAX_HW_QUEUES=%s\n", khz, T_us, ticks, M,
         getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)");
  for (int blocks : {1, 256}) {
    printf("-- %d workgroup(s) of 64 lanes per launch\n", blocks);
    for (int K : {1, 2, 3, 4, 5, 6, 8, 12, 16, 24, 32}) {
      std::vector<hipStream_t> st(K);
      for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      for (auto& s : st) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1ull, sink);   // warm every queue
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      for (int m = 0; m < M; m++)
        for (auto& s : st) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, s, ticks, sink);
      CK(hipDeviceSynchronize());
      double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6;
      printf("streams %2d: wall %9.0f us  -> %5.2f kernels active on average (ideal %d)\n", K, wall, K * M * T_us / wall, K);
      for (auto& s : st) CK(hipStreamDestroy(s));
    }
  }
  return 0;
}
