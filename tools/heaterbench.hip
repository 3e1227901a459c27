// DVFS and a lone proof: the leaf-sponge kernel takes 3.5 ms when the chip was busy just before it and 4.1 ms after
// 10 ms of idle (tools/hashbench).  Does a "heater" -- a few resident waves doing little -- keep the clock up through
// a latency-bound stretch, and what does it cost the kernel that follows?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include "poseidon.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_leaf(const u64* __restrict__ cols, size_t stride, int width, size_t n, u64* __restrict__ dig) {
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  u64 out[4];
  poseidon::hash_or_noop_strided(cols + l, stride, width, out);
  for (int i = 0; i < 4; i++) dig[4 * l + i] = out[i];
}
// one cooperative-ish chain: a single wave doing dependent work for `iters` permutations (stands for a latency-bound phase)
__global__ __launch_bounds__(64) void k_chain(u64* out, int iters) {
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = threadIdx.x + i;
  for (int r = 0; r < iters; r++) poseidon::permute(s);
  out[threadIdx.x] = s[0];
}
// heater: each wave alternates `work` multiply-adds with s_sleep until *stop != 0 (or a bounded number of rounds)
__global__ __launch_bounds__(64) void k_heater(volatile int* stop, float* sink, int work, int sleep_on, int max_rounds) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int r = 0; r < max_rounds; r++) {
    for (int i = 0; i < work; i++) a = a * b + 0.5f;
    if (sleep_on) __builtin_amdgcn_s_sleep(127);
    if (*stop) break;
  }
  if (a == 12345.f) sink[0] = a;
}

int main() {
  const size_t n = 1 << 19; const int w = 135;
  u64 *cols, *dig, *chain_out; int* stop; float* sink;
  CK(hipMalloc(&cols, n * w * 8)); CK(hipMalloc(&dig, n * 32)); CK(hipMalloc(&chain_out, 64 * 8));
  CK(hipHostMalloc(&stop, 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(cols, 0x11, n * w * 8));
  hipStream_t s_main, s_heat;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&s_main, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithPriority(&s_heat, hipStreamNonBlocking, lo));
  hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
  struct Cfg { int blocks, work, sleep_on; const char* name; };
  Cfg cfgs[] = {{0, 0, 0, "no heater"}, {1, 64, 1, "1 wave, sleeping"}, {8, 64, 1, "8 waves (1 per XCD?), sleeping"}, {256, 64, 1, "256 waves, sleeping"},
                {256, 4096, 0, "256 waves spinning"}, {1024, 64, 1, "1024 waves, sleeping"}, {1024, 4096, 0, "1024 waves spinning"}, {4096, 4096, 0, "4096 waves spinning"}};
  for (auto& c : cfgs) {
    float t_chain = 0, t_leaf = 0;
    for (int it = 0; it < 4; it++) {
      CK(hipDeviceSynchronize());
      usleep(20000);   // cold chip
      *stop = 0;
      if (c.blocks) hipLaunchKernelGGL(k_heater, dim3(c.blocks), dim3(64), 0, s_heat, stop, sink, c.work, c.sleep_on, 1 << 22);
      (void)hipEventRecord(e0, s_main);
      hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s_main, chain_out, 80);   // ~3 ms latency-bound stretch
      (void)hipEventRecord(e1, s_main);
      hipLaunchKernelGGL(k_leaf, dim3(n / 64), dim3(64), 0, s_main, cols, n, w, n, dig);
      (void)hipEventRecord(e2, s_main);
      CK(hipStreamSynchronize(s_main));
      *stop = 1;
      CK(hipDeviceSynchronize());
      float a, b; (void)hipEventElapsedTime(&a, e0, e1); (void)hipEventElapsedTime(&b, e1, e2);
      if (it) { t_chain += a; t_leaf += b; }
    }
    printf("%-34s chain of 80 permutations on one wave %.3f ms   leaf kernel after it %.3f ms\n", c.name, t_chain / 3, t_leaf / 3);
  }
  return 0;
}
