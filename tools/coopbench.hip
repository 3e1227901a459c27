// Cooperative permutations on a lone wave: correctness of every form against the per-lane permutation (device,
// random states, traces included) and chained latency per permutation.  Forms: coop.h (ds_bpermute / readlane),
// coop_lat.h (DPP rows, compiler multiply, replicated word 0 in Poseidon2's partial rounds).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "coop.h"
#include "coop_lat.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { F_P1_SHFL, F_P1_LAT, F_P1_SINGLE, F_P1_SINGLE_LAT, F_P1_TRACE_SHFL, F_P1_TRACE_LAT, F_P2_SHFL, F_P2_LAT, F_COUNT };
static const char* NAMES[] = {"poseidon  coop.h (bpermute)", "poseidon  coop_lat.h (DPP)", "poseidon  single (readlane)",
                              "poseidon  single_lat", "poseidon+trace coop.h", "poseidon+trace coop_lat.h",
                              "poseidon2+trace coop.h", "poseidon2+trace coop_lat.h"};

template <int F>
__device__ __forceinline__ u64 run_form(u64 s, int lane, const u64* k_lds, u64* tr) {
  auto emit = [&](int i, u64 v) { tr[i] = v; };
  if constexpr (F == F_P1_SHFL) return coop::poseidon_permute(s, lane, k_lds);
  if constexpr (F == F_P1_LAT) return coop::poseidon_permute_lat(s, lane, k_lds);
  if constexpr (F == F_P1_SINGLE) return coop::poseidon_permute_single(s, lane, k_lds);
  if constexpr (F == F_P1_SINGLE_LAT) return coop::poseidon_permute_single_lat(s, lane, k_lds);
  if constexpr (F == F_P1_TRACE_SHFL) return coop::poseidon_permute_trace(s, lane, k_lds, emit);
  if constexpr (F == F_P1_TRACE_LAT) return coop::poseidon_permute_trace_lat(s, lane, k_lds, emit);
  if constexpr (F == F_P2_SHFL) return coop::poseidon2_permute(s, lane, k_lds, emit);
  if constexpr (F == F_P2_LAT) return coop::poseidon2_permute_lat(s, lane, k_lds, emit);
  return 0;
}
constexpr bool is_p2(int f) { return f == F_P2_SHFL || f == F_P2_LAT; }
constexpr bool is_single(int f) { return f == F_P1_SINGLE || f == F_P1_SINGLE_LAT; }

// one block of 64 lanes = 4 groups (1 state for the single forms); states[g][12] in, out[g][12] + trace[g][106] out
template <int F>
__global__ __launch_bounds__(64) void k_check(const u64* states, u64* out, u64* trace, int reps, unsigned long long* cyc) {
  __shared__ u64 k_lds[360];
  __shared__ u64 tr_lds[4 * 106];
  if (is_p2(F)) coop::stage_poseidon2_rc(k_lds); else coop::stage_poseidon_rc(k_lds);
  const int lane = threadIdx.x, g = is_single(F) ? 0 : lane / 16, rr = is_single(F) ? lane : lane % 16;
  const size_t sb = (size_t)blockIdx.x * 4 + g;
  u64 s = rr < 12 ? states[sb * 12 + rr] : 0;
  u64* tr = tr_lds + g * 106;
  for (int i = lane; i < 4 * 106; i += 64) tr_lds[i] = 0;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) s = run_form<F>(s, lane, k_lds, tr);
  unsigned long long t1 = __builtin_readcyclecounter();
  __syncthreads();
  if (rr < 12 && (!is_single(F) || lane < 12)) out[sb * 12 + rr] = s;
  if (!is_single(F) || lane < 16)
    for (int i = rr; i < 106; i += 16) trace[sb * 106 + i] = tr[i];
  if (cyc && lane == 0) *cyc = t1 - t0;
}
// per-lane references
__global__ void k_ref(const u64* states, u64* out, u64* trace, int n, int p2, int p1trace) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
  for (int k = 0; k < 12; k++) s[k] = states[(size_t)i * 12 + k];
  u64* tr = trace + (size_t)i * 106;
  for (int k = 0; k < 106; k++) tr[k] = 0;
  if (p2) { auto e = [&](int k, u64 v) { tr[k] = v; }; poseidon2::permute_impl(s, e); }
  else if (p1trace) poseidon::permute_naive_trace(s, [&](int k, u64 v) { tr[k] = v; });
  else poseidon::permute(s);
  for (int k = 0; k < 12; k++) out[(size_t)i * 12 + k] = s[k];
}

template <int F>
static bool run(u64* d_states, int n_states) {
  const bool p2 = is_p2(F), tr = p2 || F == F_P1_TRACE_SHFL || F == F_P1_TRACE_LAT, single = is_single(F);
  u64 *d_out, *d_ref, *d_tr, *d_tr_ref; unsigned long long* d_cyc;
  CK(hipMalloc(&d_out, n_states * 96)); CK(hipMalloc(&d_ref, n_states * 96));
  CK(hipMalloc(&d_tr, n_states * 106 * 8)); CK(hipMalloc(&d_tr_ref, n_states * 106 * 8)); CK(hipMalloc(&d_cyc, 8));
  CK(hipMemset(d_tr, 0, n_states * 106 * 8));
  hipLaunchKernelGGL(k_ref, dim3((n_states + 63) / 64), dim3(64), 0, 0, d_states, d_ref, d_tr_ref, n_states, p2 ? 1 : 0, tr ? 1 : 0);
  // single forms take one state per block: blocks = n_states / 4 states checked (state index 4*b)
  hipLaunchKernelGGL(k_check<F>, dim3(n_states / 4), dim3(64), 0, 0, d_states, d_out, d_tr, 1, (unsigned long long*)nullptr);
  CK(hipDeviceSynchronize());
  std::vector<u64> o(n_states * 12), r(n_states * 12), to(n_states * 106), trr(n_states * 106);
  CK(hipMemcpy(o.data(), d_out, n_states * 96, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), d_ref, n_states * 96, hipMemcpyDeviceToHost));
  CK(hipMemcpy(to.data(), d_tr, n_states * 106 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(trr.data(), d_tr_ref, n_states * 106 * 8, hipMemcpyDeviceToHost));
  size_t bad = 0, checked = 0;
  for (int i = 0; i < n_states; i += single ? 4 : 1) {
    checked++;
    bool ok = true;
    for (int k = 0; k < 12; k++) ok &= o[(size_t)i * 12 + k] == r[(size_t)i * 12 + k];
    if (tr) for (int k = 0; k < 106; k++) ok &= to[(size_t)i * 106 + k] == trr[(size_t)i * 106 + k];
    bad += !ok;
  }
  // latency: one wave, 500 chained permutations
  hipLaunchKernelGGL(k_check<F>, dim3(1), dim3(64), 0, 0, d_states, d_out, d_tr, 20, d_cyc);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_check<F>, dim3(1), dim3(64), 0, 0, d_states, d_out, d_tr, 500, d_cyc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
  printf("%-30s %s (%zu states, %zu differ)   %7.2f us / permutation   %8.0f cycles\n", NAMES[F], bad ? "MISMATCH" : "bit-exact",
         checked, bad, ms * 1e3 / 500, (double)cyc / 500);
  hipFree(d_out); hipFree(d_ref); hipFree(d_tr); hipFree(d_tr_ref); hipFree(d_cyc);
  return bad == 0;
}

int main() {
  const int n = 4096;
  std::vector<u64> h((size_t)n * 12);
  u64 x = 0x243F6A8885A308D3ull;
  for (auto& v : h) { x += 0x9E3779B97F4A7C15ull; u64 z = x; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31; v = z % gl::P; }
  for (int k = 0; k < 12; k++) { h[k] = 0; h[12 + k] = gl::P - 1; h[24 + k] = k; }   // edge states
  u64* d; CK(hipMalloc(&d, h.size() * 8)); CK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  bool ok = true;
  ok &= run<F_P1_SHFL>(d, n); ok &= run<F_P1_LAT>(d, n); ok &= run<F_P1_SINGLE>(d, n); ok &= run<F_P1_SINGLE_LAT>(d, n);
  ok &= run<F_P1_TRACE_SHFL>(d, n); ok &= run<F_P1_TRACE_LAT>(d, n); ok &= run<F_P2_SHFL>(d, n); ok &= run<F_P2_LAT>(d, n);
  printf(ok ? "ALL FORMS BIT-EXACT\n" : "FAILED\n");
  return ok ? 0 : 1;
}
