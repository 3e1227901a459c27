// Variants of the Merkle leaf-sponge kernel on a synthetic 2^19 x 135 column-major matrix.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include "poseidon.h"
#include "poseidon_mfma.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int BS>
__global__ __launch_bounds__(BS) void k_v0(const u64* __restrict__ cols, size_t stride, int width, size_t n, u64* __restrict__ dig) {
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  u64 out[4];
  poseidon::hash_or_noop_strided(cols + l, stride, width, out);
  for (int i = 0; i < 4; i++) dig[4 * l + i] = out[i];
}
// prefetch the next 8 words while permuting
template <int BS>
__global__ __launch_bounds__(BS) void k_prefetch(const u64* __restrict__ cols, size_t stride, int width, size_t n, u64* __restrict__ dig) {
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  const u64* in = cols + l;
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = 0;
  u64 nx[8];
  for (int i = 0; i < 8; i++) nx[i] = i < width ? in[(size_t)i * stride] : 0;
  for (int off = 0; off < width; off += 8) {
    int m = width - off < 8 ? width - off : 8;
    for (int i = 0; i < 8; i++) if (i < m) s[i] = nx[i];
    int no = off + 8;
    for (int i = 0; i < 8; i++) nx[i] = (no + i) < width ? in[(size_t)(no + i) * stride] : 0;
    poseidon::permute(s);
  }
  for (int i = 0; i < 4; i++) dig[4 * l + i] = s[i];
}
// occupancy capped through dynamic LDS
template <int BS>
__global__ __launch_bounds__(BS) void k_lds(const u64* __restrict__ cols, size_t stride, int width, size_t n, u64* __restrict__ dig) {
  extern __shared__ u64 dummy[];
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  u64 out[4];
  poseidon::hash_or_noop_strided(cols + l, stride, width, out);
  if (out[0] == 0x1234567 && dummy[0] == 1) out[1] = 0;
  for (int i = 0; i < 4; i++) dig[4 * l + i] = out[i];
}
// registers only: 17 permutations per lane (ceiling)
__global__ __launch_bounds__(256) void k_regs(u64* dig, size_t n, int reps) {
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n) return;
  u64 s[12];
  for (int i = 0; i < 12; i++) s[i] = l * 12 + i;
  for (int r = 0; r < reps; r++) { s[0] ^= r; poseidon::permute(s); }
  for (int i = 0; i < 4; i++) dig[4 * l + i] = s[i];
}
// the wave-wide permutation (poseidon_mfma.h) in the leaf sponge; VAR 0: as in kernels_hash.hip, 1: inputs from
// registers instead of memory, 2: every permutation computes all rows
template <int MINW, int VAR>
__global__ __launch_bounds__(64, MINW) void k_mx(const u64* __restrict__ cols, size_t stride, int width, size_t n, u64* __restrict__ dig) {
#if defined(__HIP_DEVICE_COMPILE__)
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(threadIdx.x);
  const u64* in = cols + l;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = 0;
  for (int off = 0; off < width; off += 8) {
    const int m = width - off < 8 ? width - off : 8;
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (i < m) s[i] = VAR == 1 ? (u64)(l * 8 + off + i) : in[(size_t)(off + i) * stride];
    if (VAR == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int next = width - (off + 8);
    u32 rows = next <= 0 ? poseidon::ROWS_DIGEST : (next >= 8 ? poseidon::ROWS_CAPACITY : poseidon::ROWS_ALL);
    if (VAR == 2) rows = poseidon::ROWS_ALL;
    poseidon::mx::permute_wave(s, rows, c);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) dig[4 * l + i] = s[i];
#endif
}
// where the time of a leaf goes: cycles waiting for the 8 column loads vs cycles in the permutation (wave 0 of every 64th block)
template <int ALLROWS>
__global__ __launch_bounds__(64, 4) void k_mx_t(const u64* __restrict__ cols, size_t stride, int width, size_t n, u64* __restrict__ dig, unsigned long long* tm) {
#if defined(__HIP_DEVICE_COMPILE__)
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(threadIdx.x);
  const u64* in = cols + l;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = 0;
  unsigned long long tl = 0, tp = 0;
  for (int off = 0; off < width; off += 8) {
    const int m = width - off < 8 ? width - off : 8;
    unsigned long long c0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (i < m) s[i] = in[(size_t)(off + i) * stride];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long c1 = __builtin_readcyclecounter();
    const int next = width - (off + 8);
    u32 rows = next <= 0 ? poseidon::ROWS_DIGEST : (next >= 8 ? poseidon::ROWS_CAPACITY : poseidon::ROWS_ALL);
    if (ALLROWS) rows = poseidon::ROWS_ALL;
    poseidon::mx::permute_wave(s, rows, c);
    unsigned long long c2 = __builtin_readcyclecounter();
    tl += c1 - c0; tp += c2 - c1;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) dig[4 * l + i] = s[i];
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { tm[2 * (blockIdx.x >> 6)] = tl; tm[2 * (blockIdx.x >> 6) + 1] = tp; }
#endif
}
template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  const size_t n = 1 << 19; const int w = 135;
  u64 *cols, *dig;
  CK(hipMalloc(&cols, n * w * 8)); CK(hipMalloc(&dig, n * 32));
  CK(hipMemset(cols, 0x11, n * w * 8));
  const double perms = (double)n * 17;
  auto rep = [&](const char* name, float ms) { printf("%-28s %7.3f ms  %7.1f Mperm/s\n", name, ms, perms / ms / 1e3); };
  rep("v0 bs256", timeit([&] { hipLaunchKernelGGL(k_v0<256>, dim3(n / 256), dim3(256), 0, 0, cols, n, w, n, dig); }));
  rep("v0 bs128", timeit([&] { hipLaunchKernelGGL(k_v0<128>, dim3(n / 128), dim3(128), 0, 0, cols, n, w, n, dig); }));
  rep("v0 bs64", timeit([&] { hipLaunchKernelGGL(k_v0<64>, dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  {
    // checksum of the digests, to compare builds (-DP25_ASM_MUL=0 vs 1) bit for bit
    CK(hipMemset(cols, 0, n * w * 8));
    u64* h = (u64*)malloc(n * w * 8);
    u64 x = 88172645463325252ull;
    for (size_t i = 0; i < n * w; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = x % 0xFFFFFFFF00000001ull; }
    CK(hipMemcpy(cols, h, n * w * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_v0<64>, dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig);
    CK(hipMemcpy(h, dig, n * 32, hipMemcpyDeviceToHost));
    u64 sum = 0, xr = 0;
    for (size_t i = 0; i < n * 4; i++) { sum += h[i] * (2 * i + 1); xr ^= h[i]; }
    printf("digest checksum: %016llx %016llx\n", (unsigned long long)sum, (unsigned long long)xr);
    free(h);
  }
  rep("mx (64,4)", timeit([&] { hipLaunchKernelGGL((k_mx<4, 0>), dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  {
    u64* h = (u64*)malloc(n * 32);
    CK(hipMemcpy(h, dig, n * 32, hipMemcpyDeviceToHost));
    u64 sum = 0, xr = 0;
    for (size_t i = 0; i < n * 4; i++) { sum += h[i] * (2 * i + 1); xr ^= h[i]; }
    printf("digest checksum (mx):    %016llx %016llx\n", (unsigned long long)sum, (unsigned long long)xr);
    free(h);
  }
  rep("mx (64,3)", timeit([&] { hipLaunchKernelGGL((k_mx<3, 0>), dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  rep("mx (64,2)", timeit([&] { hipLaunchKernelGGL((k_mx<2, 0>), dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  rep("mx (64,4) inputs from registers", timeit([&] { hipLaunchKernelGGL((k_mx<4, 1>), dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  rep("mx (64,4) explicit wait after the loads", timeit([&] { hipLaunchKernelGGL((k_mx<4, 3>), dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  rep("mx (64,4) all rows", timeit([&] { hipLaunchKernelGGL((k_mx<4, 2>), dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  {
    unsigned long long* tm; CK(hipMalloc(&tm, 128 * 16));
    unsigned long long h[256];
    for (int all = 0; all < 2; all++) {
      float ms = timeit([&] { if (all) hipLaunchKernelGGL(k_mx_t<1>, dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig, tm); else hipLaunchKernelGGL(k_mx_t<0>, dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig, tm); });
      CK(hipMemcpy(h, tm, sizeof(h), hipMemcpyDeviceToHost));
      double a = 0, b = 0; for (int i = 0; i < 128; i++) { a += h[2 * i]; b += h[2 * i + 1]; }
      printf("stamped mx, allrows=%d: %.3f ms; per wave per permutation: %.0f ticks waiting for loads, %.0f ticks permuting (100 MHz ticks x %d perms)\n", all, ms, a / 128 / 17, b / 128 / 17, 17);
    }
  }
  {
    // does the leaf kernel run slower when the chip was idle just before it (a lone proof's latency-bound stretches)?
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int idle_ms : {0, 1, 3, 10, 30}) {
      float tot = 0, mx = 0;
      for (int it = 0; it < 6; it++) {
        (void)hipDeviceSynchronize();
        if (idle_ms) usleep(idle_ms * 1000);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_v0<64>, dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (it) { tot += ms; mx = ms > mx ? ms : mx; }
      }
      printf("leaf kernel after %2d ms of idle chip: %.3f ms (max %.3f)\n", idle_ms, tot / 5, mx);
    }
  }
  rep("prefetch bs256",timeit([&] { hipLaunchKernelGGL(k_prefetch<256>, dim3(n / 256), dim3(256), 0, 0, cols, n, w, n, dig); }));
  rep("prefetch bs64", timeit([&] { hipLaunchKernelGGL(k_prefetch<64>, dim3(n / 64), dim3(64), 0, 0, cols, n, w, n, dig); }));
  rep("lds 40KB bs256 (4 w/SIMD)", timeit([&] { hipLaunchKernelGGL(k_lds<256>, dim3(n / 256), dim3(256), 40 * 1024, 0, cols, n, w, n, dig); }));
  rep("lds 20KB bs256 (8 blk/CU)", timeit([&] { hipLaunchKernelGGL(k_lds<256>, dim3(n / 256), dim3(256), 20 * 1024, 0, cols, n, w, n, dig); }));
  rep("lds 54KB bs256 (3 blk/CU)", timeit([&] { hipLaunchKernelGGL(k_lds<256>, dim3(n / 256), dim3(256), 54 * 1024, 0, cols, n, w, n, dig); }));
  rep("regs only, 17 perms", timeit([&] { hipLaunchKernelGGL(k_regs, dim3(n / 256), dim3(256), 0, 0, dig, n, 17); }));
  return 0;
}
