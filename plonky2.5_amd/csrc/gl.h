// Goldilocks field F_p, p = 2^64 - 2^32 + 1, and its quadratic extension F_p[X]/(X^2 - 7).
//
// Shared by the HIP kernels (device) and the C++ host side of the product library.
// Reference semantics: GoldilocksField / QuadraticExtension of the (absent) upstream crate
// plonky2_field @ 3de92d9 as used throughout /root/reference (e.g. modulus at
// src/p3/mod.rs:55, W = 7 at src/p3/extension.rs:147-152, 2^32-th root at extension.rs:155).
//
// Representation invariant: every value stored in memory is canonical (< p) -- except what pass 1 of a two-pass NTT
// hands to pass 2 (NttPass::lazy_out).  Inside a kernel a value may be held non-canonically (any u64) between
// `mul_nc`/`reduce*` calls and through the lazy operations of gl_lazy.h; `canon` brings it back.  All arithmetic is exact integer arithmetic -> results are bit-identical on CPU and GPU.
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
#if defined(__HIP_DEVICE_COMPILE__)
GL_HD u64 make64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
// Hand-scheduled gfx950 product + reduction: 14 VALU (5 v_mad_u64_u32 + 9 carry ops) against the ~30
// the compiler emits for reduce128(a * b, mulhi64(a, b)) (it rebuilds the 128-bit product from 32-bit
// pieces through v_mov'd zero-extended pairs and resolves every wrap with cmp + 2 cndmask).
//   U + c*2^64 = a0*b1 + a1*b0;  (r3:r2:r1:r0) = a0*b0 + (U << 32) + a1*b1 * 2^64 + c*2^96
//   V + c1*2^64 = (r1:r0) + r2*(2^32-1);  W - bw*2^64 = V - r3     (2^64 = 2^32-1, 2^96 = -1 mod p)
//   result = W + (c1 - bw)*(2^32-1); neither correction can wrap (see DESIGN.md "field multiply").
// gfx940-family hazard: a VALU-written SGPR/VCC needs 2 wait states before a VALU reads it as a carry;
// inline asm is opaque to the compiler's hazard recogniser, hence the explicit s_nop.
__device__ __forceinline__ u64 mul_nc_asm(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 T, U, P0, P3, c, c1, dm, sx, sy, V;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(T), "=s"(dm) : "v"(a0), "v"(b1));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(U), "=s"(c) : "v"(a1), "v"(b0), "v"(T));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(P0), "=s"(dm) : "v"(a0), "v"(b0));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(P3), "=s"(dm) : "v"(a1), "v"(b1));
  u32 p0l = (u32)P0, p0h = (u32)(P0 >> 32), ul = (u32)U, uh = (u32)(U >> 32), p3l = (u32)P3, p3h = (u32)(P3 >> 32);
  u32 r2;
  asm("v_add_co_u32_e32 %0, vcc, %0, %3\n\ts_nop 1\n\t"         // r1 (over P0.hi)
      "v_addc_co_u32_e32 %1, vcc, %4, %5, vcc\n\ts_nop 1\n\t"   // r2
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc"                   // r3 - c (over P3.hi)
      : "+v"(p0h), "=&v"(r2), "+v"(p3h)
      : "v"(ul), "v"(p3l), "v"(uh)
      : "vcc");
  u64 lo = make64(p0l, p0h);
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(V), "=s"(c1) : "v"(r2), "v"(lo));
  u32 v0 = (u32)V, v1 = (u32)(V >> 32);
  asm("s_nop 1\n\t"
      "v_subb_co_u32_e64 %0, vcc, %0, %4, %5\n\ts_nop 1\n\t"  // W.lo = V.lo - r3' - c
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"           // W.hi ; vcc = bw
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, %6\n\t"             // + c1*(2^32-1): lo -= c1 ...
      "s_andn2_b64 %2, %6, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2\n\t"                // ... hi += c1 & ~borrow
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"               // - bw*(2^32-1): lo += bw ...
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"                 // ... hi -= bw & ~carry
      : "+v"(v0), "+v"(v1), "=&s"(sx), "=&s"(sy)
      : "v"(p3h), "s"(c), "s"(c1)
      : "vcc", "scc");  // s_andn2 writes SCC
  return make64(v0, v1);
}
// a*b + c for any u64 a, b, c (non-canonical result): the addend rides in as the 64-bit operand of the
// low partial product, whose carry-out is folded into r2 -- 16 VALU instead of multiply + canon + add.
// a*b + c <= (2^64-1)^2 + 2^64 - 1 < 2^128, so r3 cannot overflow.
__device__ __forceinline__ u64 mad_nc_asm(u64 a, u64 b, u64 addend) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 T, U, P0, P3, c, c1, k0, dm, sx, sy, V;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(T), "=s"(dm) : "v"(a0), "v"(b1));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(U), "=s"(c) : "v"(a1), "v"(b0), "v"(T));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(P0), "=s"(k0) : "v"(a0), "v"(b0), "v"(addend));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(P3), "=s"(dm) : "v"(a1), "v"(b1));
  u32 p0l = (u32)P0, p0h = (u32)(P0 >> 32), ul = (u32)U, uh = (u32)(U >> 32), p3l = (u32)P3, p3h = (u32)(P3 >> 32);
  u32 r2;
  asm("v_add_co_u32_e32 %0, vcc, %0, %3\n\ts_nop 1\n\t"         // r1 (over P0.hi)
      "v_addc_co_u32_e32 %1, vcc, %4, %5, vcc\n\ts_nop 1\n\t"   // r2
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"               // r3 - c (over P3.hi)
      "v_addc_co_u32_e64 %1, vcc, 0, %1, %6\n\ts_nop 1\n\t"     // r2 += carry of the low partial product
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc"
      : "+v"(p0h), "=&v"(r2), "+v"(p3h)
      : "v"(ul), "v"(p3l), "v"(uh), "s"(k0)
      : "vcc");
  u64 lo = make64(p0l, p0h);
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(V), "=s"(c1) : "v"(r2), "v"(lo));
  u32 v0 = (u32)V, v1 = (u32)(V >> 32);
  asm("s_nop 1\n\t"
      "v_subb_co_u32_e64 %0, vcc, %0, %4, %5\n\ts_nop 1\n\t"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, %6\n\t"
      "s_andn2_b64 %2, %6, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2\n\t"
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"
      : "+v"(v0), "+v"(v1), "=&s"(sx), "=&s"(sy)
      : "v"(p3h), "s"(c), "s"(c1)
      : "vcc", "scc");
  return make64(v0, v1);
}
// The same with a WAVE-UNIFORM second factor (a compile-time constant, a challenge, a kernel argument): its halves are
// read straight from SGPRs as the multiply-adds' scalar operand, so a 64-bit constant costs two s_mov instead of two
// v_mov per use (or two VGPRs for as long as it lives).  a, addend: any u64; non-canonical result.
__device__ __forceinline__ u64 mad_nc_s_asm(u64 a, u64 b_uniform, u64 addend) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b_uniform, b1 = (u32)(b_uniform >> 32);
  u64 T, U, P0, P3, c, c1, k0, dm, sx, sy, V;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(T), "=s"(dm) : "v"(a0), "s"(b1));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(U), "=s"(c) : "v"(a1), "s"(b0), "v"(T));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(P0), "=s"(k0) : "v"(a0), "s"(b0), "v"(addend));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(P3), "=s"(dm) : "v"(a1), "s"(b1));
  u32 p0l = (u32)P0, p0h = (u32)(P0 >> 32), ul = (u32)U, uh = (u32)(U >> 32), p3l = (u32)P3, p3h = (u32)(P3 >> 32);
  u32 r2;
  asm("v_add_co_u32_e32 %0, vcc, %0, %3\n\ts_nop 1\n\t"
      "v_addc_co_u32_e32 %1, vcc, %4, %5, vcc\n\ts_nop 1\n\t"
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
      "v_addc_co_u32_e64 %1, vcc, 0, %1, %6\n\ts_nop 1\n\t"
      "v_addc_co_u32_e32 %2, vcc, 0, %2, vcc"
      : "+v"(p0h), "=&v"(r2), "+v"(p3h)
      : "v"(ul), "v"(p3l), "v"(uh), "s"(k0)
      : "vcc");
  u64 lo = make64(p0l, p0h);
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(V), "=s"(c1) : "v"(r2), "v"(lo));
  u32 v0 = (u32)V, v1 = (u32)(V >> 32);
  asm("s_nop 1\n\t"
      "v_subb_co_u32_e64 %0, vcc, %0, %4, %5\n\ts_nop 1\n\t"
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, %6\n\t"
      "s_andn2_b64 %2, %6, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2\n\t"
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"
      : "+v"(v0), "+v"(v1), "=&s"(sx), "=&s"(sy)
      : "v"(p3h), "s"(c), "s"(c1)
      : "vcc", "scc");
  return make64(v0, v1);
}
#endif
// a*b + c, any u64 inputs -> non-canonical result
GL_HD u64 mad_nc(u64 a, u64 b, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return mad_nc_asm(a, b, c);
#else
  unsigned __int128 t = (unsigned __int128)a * b + c;
  return reduce128((u64)t, (u64)(t >> 64));
#endif
}
// a * b_uniform + c for a wave-uniform b (see mad_nc_s_asm)
GL_HD u64 mad_nc_s(u64 a, u64 b_uniform, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return mad_nc_s_asm(a, b_uniform, c);
#else
  unsigned __int128 t = (unsigned __int128)a * b_uniform + c;
  return reduce128((u64)t, (u64)(t >> 64));
#endif
}
// any u64 inputs (non-canonical allowed) -> non-canonical product
GL_HD u64 mul_nc(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return mul_nc_asm(a, b);
#else
  return reduce128(a * b, mulhi64(a, b));
#endif
}
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
