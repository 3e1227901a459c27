// Poseidon full-round MDS layer on the matrix cores (v_mfma_i32_32x32x32_i8) against the VALU form
// (poseidon::mds_rc, 24 v_mad_u64_u32 per output word), bit for bit and timed at full occupancy.
//
// Layout ("pair layout"): a wave holds 64 states; state h of batch A (h < 32) lives in the lane pair
// (h, h+32), lane h holding words {0,1,4,5,8,9} and lane h+32 words {2,3,6,7,10,11}; batch B (states
// 32..63) likewise in the other six registers.  Twelve v_permlane32_swap convert from / to the
// one-state-per-lane layout.  The state bytes ARE the B operand (K = (word, byte), 3 tiles of 32), the A
// operand is the MDS matrix expanded by a byte delta (rows = (out word, byte)), a fourth K tile injects
// the round constant and the +128 bias of the signed bytes, and the 8 partial sums of a word land in 8
// consecutive accumulator registers of the lane that owns the word.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "poseidon.h"
#include "poseidon_mfma.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

static u32 mds_entry(int i, int w) { return poseidon::MDS_CIRC[(w - i + 12) % 12] + ((i == 0 && w == 0) ? 8u : 0u); }

// [u][t][lane] -> 16 bytes
static void build_table(std::vector<int>& tbl, const u64 rc[12]) {
  tbl.assign(3 * 4 * 64 * 4, 0);
  for (int u = 0; u < 3; u++)
    for (int t = 0; t < 4; t++)
      for (int L = 0; L < 64; L++) {
        int m = L & 31, h = L >> 5;
        int r = (m & 3) + 4 * (m >> 3), hp = (m >> 2) & 1;
        int i = 4 * u + 2 * hp + (r >> 3), bp = ((r & 7) >> 1) + 4 * (r & 1);
        unsigned char by[16] = {0};
        if (t < 3) {
          for (int s = 0; s < 2; s++) by[8 * s + bp] = (unsigned char)mds_entry(i, 4 * t + 2 * h + s);
        } else {
          u32 R = 0;
          for (int w = 0; w < 12; w++) R += mds_entry(i, w);
          u32 T = 128 * R + (u32)((rc[i] >> (8 * bp)) & 255), X = T / 127, Y = T % 127;
          if (h == 0) {
            int k = 0;
            while (X) { u32 v = X < 127 ? X : 127; by[k++] = (unsigned char)v; X -= v; }
          } else by[0] = (unsigned char)Y;
        }
        int* o = &tbl[(((size_t)u * 4 + t) * 64 + L) * 4];
        for (int d = 0; d < 4; d++) o[d] = (int)((u32)by[4 * d] | ((u32)by[4 * d + 1] << 8) | ((u32)by[4 * d + 2] << 16) | ((u32)by[4 * d + 3] << 24));
      }
}

#if defined(__HIP_DEVICE_COMPILE__)
#define DEV_ONLY(...) __VA_ARGS__
#else
#define DEV_ONLY(...)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void swap_dw(u32& a, u32& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void swap_layout(u64 s[12]) {
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
      int w = 4 * t + sl;
      u32 a0 = (u32)s[w], a1 = (u32)(s[w] >> 32), b0 = (u32)s[w + 2], b1 = (u32)(s[w + 2] >> 32);
      swap_dw(a0, b0); swap_dw(a1, b1);
      s[w] = gl::make64(a0, a1); s[w + 2] = gl::make64(b0, b1);
    }
}

// 8 partial sums D_b (b = 0..7, each < 2^17, register pairs (D_k, D_{k+4})) -> sum_b D_b 2^(8b) mod p, non-canonical
__device__ __forceinline__ u64 recombine(int d0, int d1, int d2, int d3, int d4, int d5, int d6, int d7, u32 k65536) {
  u64 V0 = gl::make64((u32)d0, (u32)d1), V1 = gl::make64((u32)d2, (u32)d3), V2 = gl::make64((u32)d4, (u32)d5), V3 = gl::make64((u32)d6, (u32)d7);
  u64 AL, X, dm, t;
  // compiler-visible consumers of the accumulator (it places the MFMA -> VALU wait states; inline asm is opaque to it)
  u64 W = (V1 << 8) + V0, T = (V3 << 8) + V2;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(AL), "=s"(dm) : "v"((u32)T), "s"(k65536), "v"(W));
  u32 th = (u32)(T >> 32), ahl = th << 16, ahh = th >> 16;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(X), "=s"(dm) : "v"(ahh), "v"(AL));
  u32 x0 = (u32)X, x1 = (u32)(X >> 32);
  asm("v_add_co_u32_e32 %1, vcc, %1, %3\n\ts_nop 1\n\t"
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_addc_co_u32_e64 %1, %2, 0, %1, %2"
      : "+v"(x0), "+v"(x1), "=&s"(t) : "v"(ahl) : "vcc", "scc");
  return gl::make64(x0, x1);
}

// one batch: x[2t+sl] = register of (tile t, slot sl)
__device__ __forceinline__ void mds_mfma_batch(u64 x[6], const v4i* __restrict__ tbl, int lane, v4i bc, u32 k65536) {
  v4i b[3];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    b[t][0] = (int)((u32)x[2 * t] ^ 0x80808080u);
    b[t][1] = (int)((u32)(x[2 * t] >> 32) ^ 0x80808080u);
    b[t][2] = (int)((u32)x[2 * t + 1] ^ 0x80808080u);
    b[t][3] = (int)((u32)(x[2 * t + 1] >> 32) ^ 0x80808080u);
  }
#pragma unroll
  for (int u = 0; u < 3; u++) {
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; t++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tbl[(u * 4 + t) * 64 + lane], b[t], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tbl[(u * 4 + 3) * 64 + lane], bc, acc, 0, 0, 0);
    x[2 * u] = recombine(acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7], k65536);
    x[2 * u + 1] = recombine(acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15], k65536);
  }
}

#endif
template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_mfma(u64* st, const v4i* __restrict__ tbl, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  const int lane = threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = st[l * 12 + i];
  swap_layout(s);
  int cv = lane < 32 ? 0x7f7f7f7f : 0x01010101;
  v4i bc = {cv, cv, cv, cv};
  u32 k65536 = 65536;
  asm("" : "+s"(k65536));
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = poseidon::sbox(s[i]);
    u64 a[6] = {s[0], s[1], s[4], s[5], s[8], s[9]}, b[6] = {s[2], s[3], s[6], s[7], s[10], s[11]};
    mds_mfma_batch(a, tbl, lane, bc, k65536);
    mds_mfma_batch(b, tbl, lane, bc, k65536);
    s[0] = a[0]; s[1] = a[1]; s[4] = a[2]; s[5] = a[3]; s[8] = a[4]; s[9] = a[5];
    s[2] = b[0]; s[3] = b[1]; s[6] = b[2]; s[7] = b[3]; s[10] = b[4]; s[11] = b[5];
  }
  swap_layout(s);
#pragma unroll
  for (int i = 0; i < 12; i++) st[l * 12 + i] = gl::canon(s[i]);
#endif
}

template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_base(u64* st, int rcrow, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = st[l * 12 + i];
  poseidon::rc_ptr rc = (poseidon::rc_ptr)poseidon::RC_SPLIT.v;
  asm("" : "+s"(rc));
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = poseidon::sbox(s[i]);
    poseidon::mds_rc(s, rc + 2 * 12 * rcrow);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[l * 12 + i] = gl::canon(s[i]);
#endif
}

template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_sbox(u64* st, int iters) {
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = st[l * 12 + i];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = poseidon::sbox(s[i]);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[l * 12 + i] = gl::canon(s[i]);
}

__global__ __launch_bounds__(64) void k_dbg(const u64* st, const v4i* __restrict__ tbl, int* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = st[(size_t)lane * 12 + i];
  swap_layout(s);
#pragma unroll
  for (int i = 0; i < 12; i++) { out[(i * 64 + lane) * 2] = (int)(u32)s[i]; out[(i * 64 + lane) * 2 + 1] = (int)(u32)(s[i] >> 32); }
  int* o2 = out + 12 * 64 * 2;
  u64 x[6] = {s[0], s[1], s[4], s[5], s[8], s[9]};
  v4i b[3];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    b[t][0] = (int)((u32)x[2 * t] ^ 0x80808080u);
    b[t][1] = (int)((u32)(x[2 * t] >> 32) ^ 0x80808080u);
    b[t][2] = (int)((u32)x[2 * t + 1] ^ 0x80808080u);
    b[t][3] = (int)((u32)(x[2 * t + 1] >> 32) ^ 0x80808080u);
  }
  int cv = lane < 32 ? 0x7f7f7f7f : 0x01010101;
  v4i bc = {cv, cv, cv, cv};
  v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 3; t++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tbl[(0 * 4 + t) * 64 + lane], b[t], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tbl[(0 * 4 + 3) * 64 + lane], bc, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; r++) o2[r * 64 + lane] = acc[r];
#endif
}

// ---- the whole permutation: permute_dev (VALU) against mx::permute_wave (full-round MDS layers on the matrix cores)
template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_perm_old(u64* st, u32 rows, int reps) {
#if defined(__HIP_DEVICE_COMPILE__)
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = st[l * 12 + i];
  for (int r = 0; r < reps; r++) {
    poseidon::permute_dev(s, rows);
    if (rows == poseidon::ROWS_CAPACITY) { for (int i = 0; i < 8; i++) s[i] = s[8 + (i & 3)] ^ (u64)i; }
    if (rows == poseidon::ROWS_DIGEST) { for (int i = 4; i < 12; i++) s[i] = s[i & 3] >> 1; }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[l * 12 + i] = s[i];
#endif
}
template <int MINW>
__global__ __launch_bounds__(64, MINW) void k_perm_new(u64* st, u32 rows, int reps) {
#if defined(__HIP_DEVICE_COMPILE__)
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = st[l * 12 + i];
  const poseidon::mx::Ctx c = poseidon::mx::make_ctx(threadIdx.x);
  for (int r = 0; r < reps; r++) {
    poseidon::mx::permute_wave(s, rows, c);
    if (rows == poseidon::ROWS_CAPACITY) { for (int i = 0; i < 8; i++) s[i] = s[8 + (i & 3)] ^ (u64)i; }
    if (rows == poseidon::ROWS_DIGEST) { for (int i = 4; i < 12; i++) s[i] = s[i & 3] >> 1; }
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[l * 12 + i] = s[i];
#endif
}

__global__ __launch_bounds__(64) void k_stage(u64* st, int mode) {
#if defined(__HIP_DEVICE_COMPILE__)
  using namespace poseidon;
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = st[l * 12 + i];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = add_rc(s[i], RC[i]);
  if (mode == 0) {
    rc_ptr rc = (rc_ptr)RC_SPLIT.v;
    asm("" : "+s"(rc));
    for (int r = 1; r <= 3; r++) {
#pragma unroll
      for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
      mds_rc(s, rc + 2 * 12 * r);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
  } else if (mode == 1) {
    const mx::Ctx c = mx::make_ctx(threadIdx.x);
    u64 A[6], B[6];
    mx::to_pairs(s, A, B);
    mx::full_rounds<0, 3, true>(A, B, c, 7);
    mx::from_pairs(s, A, B);
  } else if (mode == 2) {   // one layer only, old
    rc_ptr rc = (rc_ptr)RC_SPLIT.v;
    asm("" : "+s"(rc));
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
    mds_rc(s, rc + 2 * 12 * 1);
  } else {                  // one layer only, new, no pipelining
    const mx::Ctx c = mx::make_ctx(threadIdx.x);
    u64 A[6], B[6];
    mx::to_pairs(s, A, B);
    mx::sbox6(A); mx::sbox6(B);
    mx::v16i acc[3]; mx::v4i b[3], ctile[3];
    mx::load_ctiles(ctile, c, 0);
    mx::xor_operand(A, b); mx::mfma12(acc, b, c, ctile, 7); mx::recombine6(A, acc, c);
    mx::xor_operand(B, b); mx::mfma12(acc, b, c, ctile, 7); mx::recombine6(B, acc, c);
    mx::from_pairs(s, A, B);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[l * 12 + i] = gl::canon(s[i]);
#endif
}

__global__ __launch_bounds__(64) void k_tbl(int* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  using namespace poseidon;
  const mx::Ctx c = mx::make_ctx(threadIdx.x);
  mx::v4i ctile[3];
  mx::load_ctiles(ctile, c, 0);
  for (int k = 0; k < 4; k++) for (int d = 0; d < 4; d++) out[(k * 64 + threadIdx.x) * 4 + d] = c.tk[k][d];
  for (int u = 0; u < 3; u++) for (int d = 0; d < 4; d++) out[((4 + u) * 64 + threadIdx.x) * 4 + d] = ctile[u][d];
  for (int d = 0; d < 4; d++) out[(7 * 64 + threadIdx.x) * 4 + d] = c.bc[d];
#endif
}

template <class F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  const int RCROW = 5;
  const size_t n = 1 << 19;
  u64 rc[12];
  for (int i = 0; i < 12; i++) rc[i] = poseidon::RC[12 * RCROW + i];
  std::vector<int> tbl;
  build_table(tbl, rc);
  v4i* dtbl; CK(hipMalloc(&dtbl, tbl.size() * 4)); CK(hipMemcpy(dtbl, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice));
  std::vector<u64> h0(n * 12), hb(n * 12), hm(n * 12);
  u64 x = 88172645463325252ull;
  for (size_t i = 0; i < n * 12; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h0[i] = x % gl::P; }
  // edge values in the first states
  for (int i = 0; i < 12; i++) { h0[i] = gl::P - 1; h0[12 + i] = 0; h0[24 + i] = 0xFFFFFFFFull; h0[36 + i] = gl::P - 1 - i; }
  u64 *da, *db; CK(hipMalloc(&da, n * 96)); CK(hipMalloc(&db, n * 96));
  {
    // raw accumulators of tile u = 0, batch A, against the assumed operand / result layouts
    int* dout; CK(hipMalloc(&dout, (12 * 64 * 2 + 16 * 64) * 4));
    CK(hipMemcpy(da, h0.data(), 64 * 96, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_dbg, dim3(1), dim3(64), 0, 0, da, dtbl, dout);
    CK(hipDeviceSynchronize());
    std::vector<int> ho(12 * 64 * 2 + 16 * 64);
    CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
    // expected pair layout
    size_t badsw = 0;
    for (int t = 0; t < 3; t++) for (int sl = 0; sl < 2; sl++) for (int L = 0; L < 64; L++) {
      int w = 4 * t + sl;
      u64 ea = L < 32 ? h0[(size_t)L * 12 + w] : h0[(size_t)(L - 32) * 12 + w + 2];
      u64 eb = L < 32 ? h0[(size_t)(L + 32) * 12 + w] : h0[(size_t)L * 12 + w + 2];
      u64 ga = ((u64)(u32)ho[(w * 64 + L) * 2 + 1] << 32) | (u32)ho[(w * 64 + L) * 2];
      u64 gb = ((u64)(u32)ho[((w + 2) * 64 + L) * 2 + 1] << 32) | (u32)ho[((w + 2) * 64 + L) * 2];
      if (ea != ga || eb != gb) badsw++;
    }
    printf("pair layout after the swaps: %zu mismatches\n", badsw);
    size_t badacc = 0;
    for (int L = 0; L < 64; L++) for (int r = 0; r < 16; r++) {
      int n = L & 31, hp = L >> 5;
      int i = 2 * hp + (r >> 3), bp = ((r & 7) >> 1) + 4 * (r & 1);
      long long e = 0;
      for (int w = 0; w < 12; w++) e += (long long)mds_entry(i, w) * (long long)((h0[(size_t)n * 12 + w] >> (8 * bp)) & 255);
      e += (long long)((rc[i] >> (8 * bp)) & 255);
      int g = ho[12 * 64 * 2 + r * 64 + L];
      if (e != g) { if (badacc < 6) printf("  acc lane %d reg %d: expected %lld got %d\n", L, r, e, g); badacc++; }
    }
    printf("raw accumulators (u = 0, batch A): %zu mismatches of 1024\n", badacc);
  }
  // ---- correctness: 1 and 5 rounds, against the host definition and against the VALU kernel
  for (int iters : {1, 5}) {
    CK(hipMemcpy(da, h0.data(), n * 96, hipMemcpyHostToDevice)); CK(hipMemcpy(db, h0.data(), n * 96, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_base<6>, dim3(n / 64), dim3(64), 0, 0, da, RCROW, iters);
    hipLaunchKernelGGL(k_mfma<4>, dim3(n / 64), dim3(64), 0, 0, db, dtbl, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hb.data(), da, n * 96, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm.data(), db, n * 96, hipMemcpyDeviceToHost));
    size_t bad = 0, badh = 0;
    for (size_t i = 0; i < n * 12; i++) if (hb[i] != hm[i]) { if (bad < 8) printf("  diff state %zu word %zu: valu %016llx mfma %016llx\n", i / 12, i % 12, (unsigned long long)hb[i], (unsigned long long)hm[i]); bad++; }
    for (size_t h = 0; h < 4096; h++) {
      u64 s[12];
      for (int i = 0; i < 12; i++) s[i] = h0[h * 12 + i];
      for (int it = 0; it < iters; it++) {
        for (int i = 0; i < 12; i++) s[i] = poseidon::sbox(s[i]);
        poseidon::mds(s);
        for (int i = 0; i < 12; i++) s[i] = poseidon::add_rc(s[i], rc[i]);
      }
      for (int i = 0; i < 12; i++) if (gl::canon(s[i]) != hm[h * 12 + i]) badh++;
    }
    printf("%d round(s): mfma vs valu kernel: %zu words differ of %zu; mfma vs host definition (4096 states): %zu differ\n", iters, bad, n * 12, badh);
  }
  {
    int* dout; CK(hipMalloc(&dout, 8 * 64 * 16));
    hipLaunchKernelGGL(k_tbl, dim3(1), dim3(64), 0, 0, dout);
    CK(hipDeviceSynchronize());
    std::vector<int> ho(8 * 64 * 4);
    CK(hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost));
    u64 rc1[12]; for (int i = 0; i < 12; i++) rc1[i] = poseidon::RC[12 + i];
    std::vector<int> t1; build_table(t1, rc1);
    auto kind = [](int u, int t) { return (u == 0 && t == 0) ? 3 : (t - u + 3) % 3; };
    for (int u = 0; u < 3; u++) for (int t = 0; t < 3; t++) {
      int bad = 0;
      for (int L = 0; L < 64; L++) for (int d = 0; d < 4; d++) if (t1[(((size_t)u * 4 + t) * 64 + L) * 4 + d] != ho[(kind(u, t) * 64 + L) * 4 + d]) bad++;
      printf("device state tile kind %d vs host tile (%d,%d): %d dwords differ\n", kind(u, t), u, t, bad);
    }
    for (int u = 0; u < 3; u++) {
      int bad = 0;
      for (int L = 0; L < 64; L++) for (int d = 0; d < 4; d++) if (t1[(((size_t)u * 4 + 3) * 64 + L) * 4 + d] != ho[((4 + u) * 64 + L) * 4 + d]) bad++;
      printf("device const tile layer 0 u %d: %d dwords differ\n", u, bad);
    }
    printf("bc lane0 %08x lane63 %08x\n", ho[(7 * 64 + 0) * 4], ho[(7 * 64 + 63) * 4 + 3]);
  }
  for (int pair = 0; pair < 2; pair++) {
    CK(hipMemcpy(da, h0.data(), 4096 * 96, hipMemcpyHostToDevice)); CK(hipMemcpy(db, h0.data(), 4096 * 96, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_stage, dim3(64), dim3(64), 0, 0, da, pair ? 0 : 2);
    hipLaunchKernelGGL(k_stage, dim3(64), dim3(64), 0, 0, db, pair ? 1 : 3);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hb.data(), da, 4096 * 96, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm.data(), db, 4096 * 96, hipMemcpyDeviceToHost));
    size_t bad = 0; int badw[12] = {0};
    for (size_t i = 0; i < 4096 * 12; i++) if (hb[i] != hm[i]) { bad++; badw[i % 12]++; }
    printf("%s: %zu words differ; per word:", pair ? "rounds 0-2 + sbox 3" : "one layer (round 0)", bad);
    for (int i = 0; i < 12; i++) printf(" %d", badw[i]);
    printf("\n");
  }
  // ---- whole permutation
  for (u32 rows : {poseidon::ROWS_ALL, poseidon::ROWS_DIGEST, poseidon::ROWS_CAPACITY}) {
    CK(hipMemcpy(da, h0.data(), n * 96, hipMemcpyHostToDevice)); CK(hipMemcpy(db, h0.data(), n * 96, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_perm_old<6>, dim3(n / 64), dim3(64), 0, 0, da, rows, 3);
    hipLaunchKernelGGL(k_perm_new<4>, dim3(n / 64), dim3(64), 0, 0, db, rows, 3);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(hb.data(), da, n * 96, hipMemcpyDeviceToHost)); CK(hipMemcpy(hm.data(), db, n * 96, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < n * 12; i++) {
      const int w = (int)(i % 12);
      if (!((rows >> w) & 1)) continue;
      if (hb[i] != hm[i]) { if (bad < 4) printf("  perm diff state %zu word %d: valu %016llx mfma %016llx\n", i / 12, w, (unsigned long long)hb[i], (unsigned long long)hm[i]); bad++; }
    }
    u64 s0[12]; for (int i = 0; i < 12; i++) s0[i] = h0[12 * 77 + i];
    printf("permutation x3, rows %03x: permute_wave vs permute_dev: %zu words differ\n", rows, bad);
  }
  {
    const int REPS = 17;
    const double perms = (double)n * REPS;
    auto rp = [&](const char* name, float ms) { printf("%-40s %8.3f ms  %8.1f Mperm/s\n", name, ms, perms / ms / 1e3); };
    rp("permute_dev  rows=capacity (64,6)", timeit([&] { hipLaunchKernelGGL(k_perm_old<6>, dim3(n / 64), dim3(64), 0, 0, da, poseidon::ROWS_CAPACITY, REPS); }));
    rp("permute_dev  rows=all      (64,6)", timeit([&] { hipLaunchKernelGGL(k_perm_old<6>, dim3(n / 64), dim3(64), 0, 0, da, poseidon::ROWS_ALL, REPS); }));
    rp("permute_wave rows=capacity (64,4)", timeit([&] { hipLaunchKernelGGL(k_perm_new<4>, dim3(n / 64), dim3(64), 0, 0, db, poseidon::ROWS_CAPACITY, REPS); }));
    rp("permute_wave rows=capacity (64,3)", timeit([&] { hipLaunchKernelGGL(k_perm_new<3>, dim3(n / 64), dim3(64), 0, 0, db, poseidon::ROWS_CAPACITY, REPS); }));
    rp("permute_wave rows=all      (64,4)", timeit([&] { hipLaunchKernelGGL(k_perm_new<4>, dim3(n / 64), dim3(64), 0, 0, db, poseidon::ROWS_ALL, REPS); }));
    rp("permute_wave rows=digest   (64,4)", timeit([&] { hipLaunchKernelGGL(k_perm_new<4>, dim3(n / 64), dim3(64), 0, 0, db, poseidon::ROWS_DIGEST, REPS); }));
  }
  // ---- timing
  const int IT = 64;
  auto rep = [&](const char* name, float ms, float base) { printf("%-34s %8.3f ms   %6.1f ns per wave-round   (layer part vs sbox-only: %+.3f ms)\n", name, ms, ms * 1e6 / ((double)(n / 64) * IT) * 1.0, ms - base); };
  float tsb6 = timeit([&] { hipLaunchKernelGGL(k_sbox<6>, dim3(n / 64), dim3(64), 0, 0, da, IT); });
  float tsb4 = timeit([&] { hipLaunchKernelGGL(k_sbox<4>, dim3(n / 64), dim3(64), 0, 0, da, IT); });
  rep("sbox only (64,6)", tsb6, tsb6);
  rep("sbox only (64,4)", tsb4, tsb4);
  rep("valu mds_rc (64,6)", timeit([&] { hipLaunchKernelGGL(k_base<6>, dim3(n / 64), dim3(64), 0, 0, da, RCROW, IT); }), tsb6);
  rep("valu mds_rc (64,4)", timeit([&] { hipLaunchKernelGGL(k_base<4>, dim3(n / 64), dim3(64), 0, 0, da, RCROW, IT); }), tsb4);
  rep("mfma mds (64,6)", timeit([&] { hipLaunchKernelGGL(k_mfma<6>, dim3(n / 64), dim3(64), 0, 0, db, dtbl, IT); }), tsb6);
  rep("mfma mds (64,5)", timeit([&] { hipLaunchKernelGGL(k_mfma<5>, dim3(n / 64), dim3(64), 0, 0, db, dtbl, IT); }), tsb6);
  rep("mfma mds (64,4)", timeit([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(n / 64), dim3(64), 0, 0, db, dtbl, IT); }), tsb4);
  rep("mfma mds (64,3)", timeit([&] { hipLaunchKernelGGL(k_mfma<3>, dim3(n / 64), dim3(64), 0, 0, db, dtbl, IT); }), tsb4);
  return 0;
}
