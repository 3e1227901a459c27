// Config-2 prototype: the three widest Merkle levels of a lone proof's 2^19-leaf tree (2^18, 2^17, 2^16
// parents) as the product runs them -- one per-lane launch per level -- against ONE launch in which a lane owns 8 digests and
// hashes 4 + 2 + 1 (no cross-lane traffic), and against 2 + 1 per lane followed by one ordinary level.  Idle chip, same
// inputs, outputs compared word for word.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "poseidon.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(64, 6) void k_level(const u64* __restrict__ ch, u64* __restrict__ par, size_t n) {
  size_t m = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= n) return;
  u64 l[4], r[4], o[4];
  for (int i = 0; i < 4; i++) { l[i] = ch[8 * m + i]; r[i] = ch[8 * m + 4 + i]; }
  poseidon::two_to_one(l, r, o);
  for (int i = 0; i < 4; i++) par[4 * m + i] = o[i];
}
// lane m: children 8m .. 8m+7 of level 0 -> level-1 nodes 4m..4m+3, level-2 nodes 2m, 2m+1, level-3 node m
__global__ __launch_bounds__(64, 6) void k_sub3(const u64* __restrict__ l0, u64* __restrict__ l1, u64* __restrict__ l2,
                                                u64* __restrict__ l3, size_t n3) {
  size_t m = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= n3) return;
  u64 a[4][4];
  for (int k = 0; k < 4; k++) {
    u64 l[4], r[4];
    for (int i = 0; i < 4; i++) { l[i] = l0[32 * m + 8 * k + i]; r[i] = l0[32 * m + 8 * k + 4 + i]; }
    poseidon::two_to_one(l, r, a[k]);
    for (int i = 0; i < 4; i++) l1[16 * m + 4 * k + i] = a[k][i];
  }
  u64 b[2][4];
  for (int k = 0; k < 2; k++) {
    poseidon::two_to_one(a[2 * k], a[2 * k + 1], b[k]);
    for (int i = 0; i < 4; i++) l2[8 * m + 4 * k + i] = b[k][i];
  }
  u64 c[4];
  poseidon::two_to_one(b[0], b[1], c);
  for (int i = 0; i < 4; i++) l3[4 * m + i] = c[i];
}
// lane m: children 4m .. 4m+3 -> level-1 nodes 2m, 2m+1 and level-2 node m
__global__ __launch_bounds__(64, 6) void k_sub2(const u64* __restrict__ l0, u64* __restrict__ l1, u64* __restrict__ l2, size_t n2) {
  size_t m = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= n2) return;
  u64 a[2][4];
  for (int k = 0; k < 2; k++) {
    u64 l[4], r[4];
    for (int i = 0; i < 4; i++) { l[i] = l0[16 * m + 8 * k + i]; r[i] = l0[16 * m + 8 * k + 4 + i]; }
    poseidon::two_to_one(l, r, a[k]);
    for (int i = 0; i < 4; i++) l1[8 * m + 4 * k + i] = a[k][i];
  }
  u64 c[4];
  poseidon::two_to_one(a[0], a[1], c);
  for (int i = 0; i < 4; i++) l2[4 * m + i] = c[i];
}

int main() {
  const size_t n0 = 1 << 19;
  std::vector<u64> h(4 * n0);
  u64 x = 0x243F6A8885A308D3ull;
  for (auto& v : h) { x += 0x9E3779B97F4A7C15ull; u64 z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; v = (z ^ (z >> 31)) % 0xFFFFFFFF00000001ull; }
  u64 *d0, *a1, *a2, *a3, *b1, *b2, *b3;
  CK(hipMalloc(&d0, 32 * n0));
  for (u64** p : {&a1, &b1}) CK(hipMalloc(p, 16 * n0));
  for (u64** p : {&a2, &b2}) CK(hipMalloc(p, 8 * n0));
  for (u64** p : {&a3, &b3}) CK(hipMalloc(p, 4 * n0));
  CK(hipMemcpy(d0, h.data(), 32 * n0, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](auto fn, const char* name) {
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
      hipEventRecord(e0, 0);
      fn();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it && ms < best) best = ms;
    }
    printf("%-58s %8.1f us\n", name, best * 1e3f);
    return 0;
  };
  timed([&] {
    hipLaunchKernelGGL(k_level, dim3(n0 / 2 / 64), dim3(64), 0, 0, d0, a1, n0 / 2);
    hipLaunchKernelGGL(k_level, dim3(n0 / 4 / 64), dim3(64), 0, 0, a1, a2, n0 / 4);
    hipLaunchKernelGGL(k_level, dim3(n0 / 8 / 64), dim3(64), 0, 0, a2, a3, n0 / 8);
  }, "three per-lane launches (2^18, 2^17, 2^16 parents)");
  timed([&] { hipLaunchKernelGGL(k_sub3, dim3(n0 / 8 / 64), dim3(64), 0, 0, d0, b1, b2, b3, n0 / 8); },
        "one launch, 4 + 2 + 1 hashes per lane (2^16 lanes)");
  std::vector<u64> ra(4 * (n0 / 8)), rb(4 * (n0 / 8));
  CK(hipMemcpy(ra.data(), a3, ra.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(rb.data(), b3, rb.size() * 8, hipMemcpyDeviceToHost));
  size_t diff = 0;
  for (size_t i = 0; i < ra.size(); i++) diff += ra[i] != rb[i];
  printf("   level-3 nodes: %zu of %zu words differ\n", diff, ra.size());
  CK(hipMemset(b2, 0, 8 * n0)); CK(hipMemset(b3, 0, 4 * n0));
  timed([&] {
    hipLaunchKernelGGL(k_sub2, dim3(n0 / 4 / 64), dim3(64), 0, 0, d0, b1, b2, n0 / 4);
    hipLaunchKernelGGL(k_level, dim3(n0 / 8 / 64), dim3(64), 0, 0, b2, b3, n0 / 8);
  }, "2 + 1 hashes per lane (2^17 lanes), then one ordinary level");
  CK(hipMemcpy(rb.data(), b3, rb.size() * 8, hipMemcpyDeviceToHost));
  diff = 0;
  for (size_t i = 0; i < ra.size(); i++) diff += ra[i] != rb[i];
  printf("   level-3 nodes: %zu of %zu words differ\n", diff, ra.size());
  // the next three levels (2^15, 2^14, 2^13 parents): per-lane launches against one 4 + 2 + 1 launch of 2^13 lanes
  timed([&] {
    hipLaunchKernelGGL(k_level, dim3(n0 / 16 / 64), dim3(64), 0, 0, a3, a1, n0 / 16);
    hipLaunchKernelGGL(k_level, dim3(n0 / 32 / 64), dim3(64), 0, 0, a1, a2, n0 / 32);
    hipLaunchKernelGGL(k_level, dim3(n0 / 64 / 64), dim3(64), 0, 0, a2, b3, n0 / 64);
  }, "three per-lane launches (2^15, 2^14, 2^13 parents)");
  timed([&] { hipLaunchKernelGGL(k_sub3, dim3(n0 / 64 / 64), dim3(64), 0, 0, a3, b1, b2, b3, n0 / 64); },
        "one launch, 4 + 2 + 1 hashes per lane (2^13 lanes)");
  return 0;
}
