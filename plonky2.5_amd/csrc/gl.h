// Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension F_p[X]/(X^2 - 7).
//
// Shared by the HIP kernels (device) and the C++ host side of the product library.
// Reference semantics: GoldilocksField / QuadraticExtension of the (absent) upstream crate
// plonky2_field @ 3de92d9 as used throughout /root/reference (e.g. modulus at
// src/p3/mod.rs:55, W = 7 at src/p3/extension.rs:147-152, 2^32-th root at extension.rs:155).
//
// Representation invariant: every value stored in memory is canonical (< p).  Inside a kernel a
// value may be held non-canonically (any u64) between `mul_nc`/`reduce*` calls; `canon` brings it
// back.  All arithmetic is exact integer arithmetic -> results are bit-identical on CPU and GPU.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

typedef uint64_t u64;
typedef uint32_t u32;

namespace gl {

constexpr u64 P = 0xFFFFFFFF00000001ULL;
constexpr u64 EPS = 0xFFFFFFFFULL;  // 2^64 mod p
constexpr u64 GENERATOR = 7;        // multiplicative generator, also the LDE coset shift
constexpr u64 ROOT_2_32 = 1753635133440165772ULL;  // primitive 2^32-th root of unity
constexpr u64 EXT_W = 7;            // x^2 = 7

GL_HD u64 canon(u64 x) { return x >= P ? x - P : x; }

// a, b canonical -> canonical
GL_HD u64 add(u64 a, u64 b) {
  u64 s = a + b;
  bool over = s < a;
  u64 t = s - P;  // wrapping
  return (over || s >= P) ? t : s;
}
GL_HD u64 sub(u64 a, u64 b) {
  u64 d = a - b;
  return a < b ? d + P : d;
}
GL_HD u64 neg(u64 a) { return a ? P - a : 0; }

GL_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

// (hi:lo) mod p, result possibly non-canonical (any u64 congruent to the input).
GL_HD u64 reduce128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= EPS;
  u64 t1 = hi_lo * EPS;  // < 2^64
  u64 t2 = t0 + t1;
  if (t2 < t1) t2 += EPS;
  return t2;
}
// lo + hi32 * 2^64 mod p, non-canonical result.
GL_HD u64 reduce96(u64 lo, u32 hi) {
  u64 t1 = (u64)hi * EPS;
  u64 t2 = lo + t1;
  if (t2 < t1) t2 += EPS;
  return t2;
}
// any u64 inputs (non-canonical allowed) -> non-canonical product
GL_HD u64 mul_nc(u64 a, u64 b) { return reduce128(a * b, mulhi64(a, b)); }
GL_HD u64 mul(u64 a, u64 b) { return canon(mul_nc(a, b)); }
GL_HD u64 sqr(u64 a) { return mul(a, a); }
// a*b + c, all canonical
GL_HD u64 mad(u64 a, u64 b, u64 c) { return add(mul(a, b), c); }

GL_HD u64 pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = mul(r, b);
    b = mul(b, b);
    e >>= 1;
  }
  return r;
}
GL_HD u64 inv(u64 a) { return pow(a, P - 2); }
GL_HD u64 exp_pow2(u64 b, unsigned k) {
  for (unsigned i = 0; i < k; i++) b = mul(b, b);
  return b;
}
// primitive 2^k-th root of unity (k <= 32)
GL_HD u64 root_of_unity(unsigned k) { return exp_pow2(ROOT_2_32, 32 - k); }

// ---- quadratic extension ----
struct E2 {
  u64 a, b;  // a + b X
};
GL_HD E2 e2(u64 a, u64 b = 0) { return E2{a, b}; }
GL_HD bool eq(E2 x, E2 y) { return x.a == y.a && x.b == y.b; }
GL_HD E2 add(E2 x, E2 y) { return E2{add(x.a, y.a), add(x.b, y.b)}; }
GL_HD E2 sub(E2 x, E2 y) { return E2{sub(x.a, y.a), sub(x.b, y.b)}; }
GL_HD E2 neg(E2 x) { return E2{neg(x.a), neg(x.b)}; }
GL_HD E2 mul(E2 x, E2 y) {
  u64 a = add(mul(x.a, y.a), mul(EXT_W, mul(x.b, y.b)));
  u64 b = add(mul(x.a, y.b), mul(x.b, y.a));
  return E2{a, b};
}
GL_HD E2 mul(E2 x, u64 s) { return E2{mul(x.a, s), mul(x.b, s)}; }
GL_HD E2 sqr(E2 x) { return mul(x, x); }
GL_HD E2 inv(E2 x) {
  u64 n = sub(mul(x.a, x.a), mul(EXT_W, mul(x.b, x.b)));
  u64 ni = inv(n);
  return E2{mul(x.a, ni), mul(neg(x.b), ni)};
}
GL_HD E2 pow(E2 b, u64 e) {
  E2 r = e2(1);
  while (e) {
    if (e & 1) r = mul(r, b);
    b = mul(b, b);
    e >>= 1;
  }
  return r;
}
GL_HD E2 exp_pow2(E2 b, unsigned k) {
  for (unsigned i = 0; i < k; i++) b = mul(b, b);
  return b;
}

GL_HD u32 bitrev(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  u32 r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}

}  // namespace gl
