// Cross-checks the hand-written gfx950 sequences in gl.h / poseidon.h (mul_nc_asm, mds_rc, permute_dev)
// against the plain C++ forms of the same functions, on the device, over edge cases and random inputs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "poseidon.h"
#include "ntt16.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_mul(const u64* a, const u64* b, u64* out_asm, u64* out_c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  if (i >= n) return;
  out_asm[i] = gl::canon(gl::mul_nc_asm(a[i], b[i]));
  out_c[i] = gl::canon(gl::reduce128(a[i] * b[i], gl::mulhi64(a[i], b[i])));
#endif
}
__global__ void k_mad(const u64* a, const u64* b, u64* out_asm, u64* out_c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  if (i >= n) return;
  const u64 c = a[(i * 7 + 3) % n] ^ b[(i * 5 + 1) % n];  // any u64
  out_asm[i] = gl::canon(gl::mad_nc_asm(a[i], b[i], c));
  unsigned __int128 t = (unsigned __int128)a[i] * b[i] + c;
  out_c[i] = gl::canon(gl::reduce128((u64)t, (u64)(t >> 64)));
#endif
}
__global__ void k_mds(const u64* in, u64* out_asm, u64* out_c, size_t n, int row) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  if (i >= n) return;
  u64 s[12], t[12];
  for (int k = 0; k < 12; k++) s[k] = t[k] = in[i * 12 + k];
  poseidon::mds_rc(s, (poseidon::rc_ptr)poseidon::RC_SPLIT.v + 24 * row);
  poseidon::mds(t);
  for (int k = 0; k < 12; k++) {
    out_asm[i * 12 + k] = gl::canon(s[k]);
    u64 c = row < 30 ? poseidon::RC[12 * row + k] : 0;
    out_c[i * 12 + k] = gl::add(gl::canon(t[k]), c);
  }
#endif
}

// ---- lazy arithmetic of the NTT butterflies (gl_lazy.h) against its C++ definition, and both against canonical arithmetic
__global__ void k_bfly(const u64* a, const u64* b, u64* out_asm, u64* out_c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  if (i >= n) return;
  u64 s, d, s2, d2;
  gl::bfly_nc_asm(a[i], b[i], false, s, d);
  gl::bfly_nc_asm(a[i], b[i], true, s2, d2);
  out_asm[4 * i] = gl::canon(s); out_asm[4 * i + 1] = gl::canon(d); out_asm[4 * i + 2] = gl::canon(s2); out_asm[4 * i + 3] = gl::canon(d2);
  const u64 ca = gl::canon(a[i]), cb = gl::canon(b[i]);
  out_c[4 * i] = gl::add(ca, cb); out_c[4 * i + 1] = gl::sub(ca, cb); out_c[4 * i + 2] = gl::add(ca, cb); out_c[4 * i + 3] = gl::sub(cb, ca);
  // the C++ definition picks the same representative or at least the same element
  if (gl::canon(gl::add_nc_c(a[i], b[i])) != out_c[4 * i] || gl::canon(gl::sub_nc_c(a[i], b[i])) != out_c[4 * i + 1]) out_c[4 * i] ^= 1;
#endif
}
template <int E0>
__global__ void k_shl(const u64* a, u64* out_asm, u64* out_c, size_t n) {   // exponents E0 .. E0 + 7, compile-time each
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  if (i >= n) return;
  const u64 x = a[i], cx = gl::canon(x);
#pragma unroll
  for (int k = 0; k < 8; k++) {
    constexpr int dummy = 0; (void)dummy;
    const int e = E0 + k;
    out_asm[8 * i + k] = gl::canon(gl::shl_nc_asm(x, e));
    u64 r = cx;                                   // x * 2^e by e canonical doublings
    for (int j = 0; j < e; j++) r = gl::add(r, r);
    if (gl::canon(gl::shl_nc_c(x, e)) != r) r ^= 1;
    out_c[8 * i + k] = r;
  }
#endif
}
template <bool INV>
__global__ void k_dft16(const u64* a, u64* out_fast, u64* out_naive, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 x[16], c[16];
  for (int k = 0; k < 16; k++) { x[k] = a[16 * i + k]; c[k] = gl::canon(x[k]); }
  p25::dft16<INV>(x);
  u64 w = gl::root_of_unity(4);
  if (INV) w = gl::inv(w);
  for (int j = 0; j < 16; j++) {
    u64 acc = 0, wj = gl::pow(w, j), t = 1;
    for (int k = 0; k < 16; k++) { acc = gl::add(acc, gl::mul(c[k], t)); t = gl::mul(t, wj); }
    const int pos = ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3);   // frequency j sits at rev4(j)
    out_naive[16 * i + pos] = acc;
  }
  for (int k = 0; k < 16; k++) out_fast[16 * i + k] = gl::canon(x[k]);
}

// single lazy operations of the quotient's evaluators: any + any, any - any, any +/- (b <= p), the scalar-operand multiply-add,
// and the wrapped decrements inside vanishing products (canonical a)
__global__ void k_single(const u64* a, const u64* b, u64* out_asm, u64* out_c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
  if (i >= n) return;
  const u64 x = a[i], y = b[i], cx = gl::canon(x), cy = gl::canon(y);
  const u64 yb = y > gl::P ? gl::P : y;          // a legal second operand of the single-correction forms (<= p)
  u64* o = out_asm + 8 * i; u64* r = out_c + 8 * i;
  o[0] = gl::canon(gl::add_nc_asm(x, y));  r[0] = gl::add(cx, cy);
  o[1] = gl::canon(gl::sub_nc_asm(x, y));  r[1] = gl::sub(cx, cy);
  o[2] = gl::canon(gl::add_c_asm(x, yb));  r[2] = gl::add(cx, gl::canon(yb));
  o[3] = gl::canon(gl::sub_c_asm(x, yb));  r[3] = gl::sub(cx, gl::canon(yb));
  // wave-uniform factor: the block's first operand, made uniform per wave
  const u64 ku = b[(i / 64) * 64];
  o[4] = gl::canon(gl::mad_nc_s_asm(x, ku, y));  r[4] = gl::add(gl::mul(cx, gl::canon(ku)), cy);
  // vanishing products on canonical values, small ones included
  const u64 l = (i & 1) ? (cx & 7) : cx;
  o[5] = gl::canon(gl::mul_nc(l, gl::dec_wrap(l, 1)));  r[5] = gl::mul(l, gl::sub(l, 1));
  o[6] = gl::canon(gl::mul_nc(gl::mul_nc(l, gl::dec_wrap(l, 1)), gl::mul_nc(gl::dec_wrap(l, 2), gl::dec_wrap(l, 3))));
  r[6] = gl::mul(gl::mul(l, gl::sub(l, 1)), gl::mul(gl::sub(l, 2), gl::sub(l, 3)));
  o[7] = gl::canon(gl::sub_c_asm(y, 1) );  r[7] = gl::sub(cy, 1);
#endif
}
__global__ void k_dump(u64* out) {
#if defined(__HIP_DEVICE_COMPILE__)
  for (int i = 0; i < 48; i++) out[i] = poseidon::RC_SPLIT.v[696 + i];
#endif
}
__device__ void permute_c(u64 s[12]) {
  using namespace poseidon;
  for (int r = 0; r < 30; r++) {
    for (int i = 0; i < 12; i++) s[i] = gl::add(gl::canon(s[i]), RC[12 * r + i]);
    if (r < 4 || r >= 26) {
      for (int i = 0; i < 12; i++) {
        u64 x = s[i], x2 = gl::canon(gl::reduce128(x * x, gl::mulhi64(x, x)));
        u64 x4 = gl::canon(gl::reduce128(x2 * x2, gl::mulhi64(x2, x2)));
        u64 x3 = gl::canon(gl::reduce128(x * x2, gl::mulhi64(x, x2)));
        s[i] = gl::canon(gl::reduce128(x3 * x4, gl::mulhi64(x3, x4)));
      }
    } else {
      u64 x = s[0], x2 = gl::canon(gl::reduce128(x * x, gl::mulhi64(x, x)));
      u64 x4 = gl::canon(gl::reduce128(x2 * x2, gl::mulhi64(x2, x2)));
      u64 x3 = gl::canon(gl::reduce128(x * x2, gl::mulhi64(x, x2)));
      s[0] = gl::canon(gl::reduce128(x3 * x4, gl::mulhi64(x3, x4)));
    }
    mds(s);
  }
  for (int i = 0; i < 12; i++) s[i] = gl::canon(s[i]);
}
__global__ void k_perm(const u64* in, u64* out_asm, u64* out_c, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12], t[12];
  for (int k = 0; k < 12; k++) s[k] = t[k] = in[i * 12 + k];
  poseidon::permute(s);
  permute_c(t);
  for (int k = 0; k < 12; k++) {
    out_asm[i * 12 + k] = s[k];
    out_c[i * 12 + k] = t[k];
  }
}

static u64 rng_state = 88172645463325252ull;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
  const size_t n = 1 << 20;
  u64 *a, *b, *o1, *o2;
  CK(hipMallocManaged(&a, n * 12 * 8)); CK(hipMallocManaged(&b, n * 8));
  CK(hipMallocManaged(&o1, n * 12 * 8)); CK(hipMallocManaged(&o2, n * 12 * 8));
  const u64 P = 0xFFFFFFFF00000001ull;
  const u64 edge[] = {0, 1, 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, P - 1, P, P + 1, ~0ull, ~0ull - 1,
                      0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull, 0x00000001FFFFFFFFull, 0xFFFFFFFF00000002ull};
  const int ne = sizeof(edge) / 8;
  int bad_total = 0;
  {
    size_t k = 0;
    for (int i = 0; i < ne; i++) for (int j = 0; j < ne; j++) { a[k] = edge[i]; b[k] = edge[j]; k++; }
    for (; k < n; k++) {
      a[k] = rnd(); b[k] = rnd();
      if ((k & 7) == 1) a[k] |= 0xFFFFFFFF00000000ull;
      if ((k & 7) == 2) b[k] &= 0xFFFFFFFFull;
      if ((k & 7) == 3) { a[k] |= 0xFFFFFFFF00000000ull; b[k] |= 0xFFFFFFFF00000000ull; }
      if ((k & 7) == 4) { a[k] = ~0ull - (rnd() & 0xFFFF); b[k] = ~0ull - (rnd() & 0xFFFF); }
    }
    hipLaunchKernelGGL(k_mul, dim3(n / 256), dim3(256), 0, 0, a, b, o1, o2, n);
    CK(hipDeviceSynchronize());
    int bad = 0;
    for (size_t i = 0; i < n; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("mul mismatch a=%016llx b=%016llx asm=%016llx c=%016llx\n", (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    printf("mul_nc_asm: %d mismatches of %zu\n", bad, n);
    bad_total += bad;
    // a*b + c on the same operands (the edge-case block makes a, b, c all-ones etc. meet)
    for (size_t i = 0; i < 64; i++) { a[n - 1 - i] = ~0ull; b[n - 1 - i] = ~0ull - (i & 1); }
    hipLaunchKernelGGL(k_mad, dim3(n / 256), dim3(256), 0, 0, a, b, o1, o2, n);
    CK(hipDeviceSynchronize());
    bad = 0;
    for (size_t i = 0; i < n; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("mad mismatch a=%016llx b=%016llx asm=%016llx c=%016llx\n", (unsigned long long)a[i], (unsigned long long)b[i], (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    printf("mad_nc_asm: %d mismatches of %zu\n", bad, n);
    bad_total += bad;
  }

  // ---- lazy butterflies: all pairs of boundary values (the double wraps live there), then random operands biased to the top
  {
    const u64 edge2[] = {0, 1, 2, 0xFFFFFFFEull, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, 0xFFFFFFFF00000000ull, P - 2, P - 1, P, P + 1, P + 2,
                         ~0ull, ~0ull - 1, ~0ull - 2, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull, 0xFFFFFFFF00000002ull,
                         0xFFFFFFFFFFFF0000ull, 0xFFFFFFFF0000FFFFull, 0x00000000FFFF0000ull, 0xFFFFFFFE00000000ull};
    const int ne2 = sizeof(edge2) / 8;
    size_t k = 0;
    for (int i = 0; i < ne2; i++) for (int j = 0; j < ne2; j++) { a[k] = edge2[i]; b[k] = edge2[j]; k++; }
    const size_t nb = 1 << 18;
    for (; k < nb; k++) {
      a[k] = rnd(); b[k] = rnd();
      if ((k & 7) == 1) { a[k] |= 0xFFFFFFFF00000000ull; b[k] |= 0xFFFFFFFF00000000ull; }
      if ((k & 7) == 2) { a[k] = ~0ull - (rnd() & 0xFFFFFFFFull); b[k] = ~0ull - (rnd() & 0xFFFFFFFFull); }
      if ((k & 7) == 3) { a[k] &= 0xFFFFFFFFull; b[k] = ~0ull - (rnd() & 0x1FFFFFFFFull); }
      if ((k & 7) == 4) { b[k] &= 0xFFFFFFFFull; a[k] = ~0ull - (rnd() & 0x1FFFFFFFFull); }
      if ((k & 7) == 5) { a[k] = rnd() & 0x1FFFFFFFFull; b[k] = a[k] + (rnd() & 3) - 1; }
    }
    hipLaunchKernelGGL(k_bfly, dim3(nb / 256), dim3(256), 0, 0, a, b, o1, o2, nb);
    CK(hipDeviceSynchronize());
    int bad = 0;
    for (size_t i = 0; i < nb * 4; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("bfly mismatch a=%016llx b=%016llx out %d asm=%016llx c=%016llx\n", (unsigned long long)a[i / 4], (unsigned long long)b[i / 4], (int)(i & 3), (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    printf("bfly_nc_asm: %d mismatches of %zu\n", bad, nb * 4);
    bad_total += bad;
    hipLaunchKernelGGL(k_single, dim3(nb / 256), dim3(256), 0, 0, a, b, o1, o2, nb);
    CK(hipDeviceSynchronize());
    bad = 0;
    for (size_t i = 0; i < nb * 8; i++) if (o1[i] != o2[i]) { if (bad < 8) printf("single-op mismatch a=%016llx b=%016llx op %d asm=%016llx c=%016llx\n", (unsigned long long)a[i / 8], (unsigned long long)b[i / 8], (int)(i & 7), (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    printf("add_nc / sub_nc / add_c / sub_c / mad_nc_s / vanishing products: %d mismatches of %zu\n", bad, nb * 8);
    bad_total += bad;
    // shifts: every exponent 0..95 on the same operand list
#define SHL_RUN(E0) hipLaunchKernelGGL(k_shl<E0>, dim3(nb / 256), dim3(256), 0, 0, a, o1, o2, nb); CK(hipDeviceSynchronize()); \
    for (size_t i = 0; i < nb * 8; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("shl mismatch x=%016llx e=%d asm=%016llx c=%016llx\n", (unsigned long long)a[i / 8], (int)(E0 + (i & 7)), (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    bad = 0;
    SHL_RUN(0) SHL_RUN(8) SHL_RUN(16) SHL_RUN(24) SHL_RUN(32) SHL_RUN(40) SHL_RUN(48) SHL_RUN(56) SHL_RUN(64) SHL_RUN(72) SHL_RUN(80) SHL_RUN(88)
    printf("shl_nc_asm (e = 0..95): %d mismatches of %zu\n", bad, nb * 8 * 12);
    bad_total += bad;
    // the 16-point network against the naive DFT (operands: the same list, sixteen at a time)
    const size_t nd = nb / 16;
    hipLaunchKernelGGL(k_dft16<false>, dim3(nd / 64), dim3(64), 0, 0, a, o1, o2, nd);
    CK(hipDeviceSynchronize());
    bad = 0;
    for (size_t i = 0; i < nd * 16; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("dft16 fwd mismatch at %zu: %016llx vs %016llx\n", i, (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    hipLaunchKernelGGL(k_dft16<true>, dim3(nd / 64), dim3(64), 0, 0, b, o1, o2, nd);
    CK(hipDeviceSynchronize());
    for (size_t i = 0; i < nd * 16; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("dft16 inv mismatch at %zu: %016llx vs %016llx\n", i, (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    printf("dft16 (lazy, fwd + inv) vs naive DFT: %d mismatches of %zu\n", bad, nd * 32);
    bad_total += bad;
  }
  const size_t m = 1 << 16;
  for (int row = 0; row <= 30; row += 5) {
    for (size_t i = 0; i < m * 12; i++) { a[i] = rnd(); if ((i % 5) == 0) a[i] |= 0xFFFFFFFF00000000ull; if ((i % 7) == 0) a[i] |= 0xFFFFFFFFull; }
    for (int i = 0; i < 24; i++) a[i] = ~0ull;
    hipLaunchKernelGGL(k_mds, dim3(m / 256), dim3(256), 0, 0, a, o1, o2, m, row);
    CK(hipDeviceSynchronize());
    int bad = 0;
    for (size_t i = 0; i < m * 12; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("mds mismatch at %zu: asm=%016llx c=%016llx\n", i, (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    printf("mds_rc row %d: %d mismatches of %zu\n", row, bad, m * 12);
    bad_total += bad;
  }
  {
    for (size_t i = 0; i < m * 12; i++) a[i] = rnd() % P;
    for (int i = 0; i < 12; i++) a[i] = i;
    hipLaunchKernelGGL(k_perm, dim3(m / 64), dim3(64), 0, 0, a, o1, o2, m);
    CK(hipDeviceSynchronize());
    int bad = 0;
    for (size_t i = 0; i < m * 12; i++) if (o1[i] != o2[i]) { if (bad < 5) printf("perm mismatch at %zu: asm=%016llx c=%016llx\n", i, (unsigned long long)o1[i], (unsigned long long)o2[i]); bad++; }
    printf("permute: %d mismatches of %zu ; perm(0..11)[0] = %016llx\n", bad, m * 12, (unsigned long long)o1[0]);
    bad_total += bad;
  }
  printf(bad_total ? "ASMCHECK FAIL\n" : "ASMCHECK OK\n");
  return bad_total ? 1 : 0;
}
