// Lazy (non-canonical) Goldilocks arithmetic: every operand and every result is ANY u64 congruent to the field element it
// stands for; `gl::canon` brings a value back before it leaves the kernel.  Used by the NTT butterflies (ntt16.h,
// kernels_ntt.hip), the quotient's gate evaluators (kernels_quotient.hip) and the openings / FRI / partial-product kernels.
//
// Why: the compiler's canonical forms cost 8 VALU instructions for gl::add (64-bit add, two 64-bit compares, a subtract, two
// selects), 6 for gl::sub, 22 for a product by a power of two and 4 for every canonical step after a multiply -- a canonical
// radix-2 butterfly on a power-of-two twiddle is 37.  The sequences below are 12 for a sum and a difference together, 4 / 6
// for a single add or subtract (second operand canonical / any) and 6 / 10 / 8 for a product by a power of two, with no
// compare and no select: a wrap of the 64-bit register is worth 2^64 = 2^32 - 1 (mod p), so every correction is "low word
// -= carry, high word += carry unless the low word borrowed", driven by the carry bit itself.  A SECOND wrap is possible
// (a + b >= 2^64 + p, one pair in 2^32 at random) and is corrected the same way; a third is not (proofs at each function).
//
// The plain C++ forms (`*_c`) are the definition: tests/native/lazy_defs.cpp checks them against 128-bit integer arithmetic
// on the CPU, tools/asmcheck.hip compares the gfx950 sequences with them and with canonical arithmetic on the device (all
// pairs of 24 boundary values, random operands biased to the top of the range, every shift exponent).
//
// gfx940-family hazard (as in gl.h): a VALU-written SGPR / VCC needs two wait states before a VALU reads it as a carry, and
// the inside of an asm block is opaque to the compiler's hazard recogniser -- hence the interleaving of the two carry chains
// of a butterfly and the explicit s_nop between dependent steps.  (A timing-only build without them bounds their cost at
// <= 0.3 % of the pipeline: profiles/r05_v_hazard_nops_bound.txt.)
#pragma once
#include <cassert>
#include "gl.h"

namespace gl {

// ---- definitions (any u64 in, any u64 out) --------------------------------------------------------------------------
// a + b = s + c * 2^64 = s + c * EPS.  If c: s <= 2^64 - 2, t = s + EPS wraps only when s >= p, and then
// t' = s + EPS - 2^64 < EPS, so t' + EPS < 2^33 cannot wrap again.
GL_HD u64 add_nc_c(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a) {
    u64 t = s + EPS;
    if (t < s) t += EPS;
    s = t;
  }
  return s;
}
// a - b = d - bw * 2^64 = d - bw * EPS.  If bw and d < EPS the subtraction borrows again: d - EPS + 2^64 >= 2^64 - EPS
// >= EPS, so the second d - EPS cannot borrow.
GL_HD u64 sub_nc_c(u64 a, u64 b) {
  u64 d = a - b;
  if (a < b) {
    u64 t = d - EPS;
    if (d < EPS) t -= EPS;
    d = t;
  }
  return d;
}
// The single-correction forms: the SECOND operand is at most p (a canonical value, or p itself).
// a + b with a carry: s = a + b - 2^64 <= p - 1, so s + EPS <= 2^64 - 1 cannot wrap.
GL_HD u64 add_c_c(u64 a, u64 b) {
  u64 s = a + b;
  if (s < a) s += EPS;
  return s;
}
// a - b with a borrow: a < b <= p, d = a - b + 2^64 >= 2^64 - p = EPS, so d - EPS cannot borrow.
GL_HD u64 sub_c_c(u64 a, u64 b) {
  u64 d = a - b;
  if (a < b) d -= EPS;
  return d;
}
// x * 2^e, 0 <= e < 96 (2^96 = -1 mod p: larger exponents are a sign, absorbed by the caller)
GL_HD u64 shl_nc_c(u64 x, int e) {
  if (e == 0) return x;
  if (e < 64) return reduce128(x << e, x >> (64 - e));
  return reduce128(x * (EPS << (e - 64)), mulhi64(x, EPS << (e - 64)));   // 2^64 = EPS
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 sequences -----------------------------------------------------------------------------------------------
// lo + r2 * 2^64 - r3 * 2^96... i.e. (r3 : r2 : lo) mod p for 32-bit r2, r3 and any 64-bit lo: the tail of mul_nc_asm.
//   V + c1*2^64 = lo + r2*(2^32-1);  W - bw*2^64 = V - r3;  result = W + (c1 - bw)*(2^32-1).
// c1 = 1 => V <= 2^64 - 2^33, so W + EPS cannot wrap; bw = 1 => W > 2^64 - 2^32, so W - EPS cannot borrow.
// 7 VALU + 2 SALU.
__device__ __forceinline__ u64 fold128_asm(u64 lo, u32 r2, u32 r3) {
  u64 V, c1, sx, sy;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(V), "=s"(c1) : "v"(r2), "v"(lo));
  u32 v0 = (u32)V, v1 = (u32)(V >> 32);
  asm("v_sub_co_u32_e32 %0, vcc, %0, %4\n\ts_nop 1\n\t"      // W.lo = V.lo - r3
      "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"          // W.hi ; vcc = bw
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, %5\n\t"            // + c1*(2^32-1): lo -= c1 ...
      "s_andn2_b64 %2, %5, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2\n\t"               // ... hi += c1 & ~borrow
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"              // - bw*(2^32-1): lo += bw ...
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"                // ... hi -= bw & ~carry
      : "+v"(v0), "+v"(v1), "=&s"(sx), "=&s"(sy)
      : "v"(r3), "s"(c1)
      : "vcc", "scc");
  return make64(v0, v1);
}
// lo + r2 * 2^64 mod p (r2 32-bit): one multiply-add and one correction, 3 VALU + 1 SALU.
__device__ __forceinline__ u64 fold96_asm(u64 lo, u32 r2) {
  u64 V, c1, sx, sy;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(V), "=s"(c1) : "v"(r2), "v"(lo));
  u32 v0 = (u32)V, v1 = (u32)(V >> 32);
  asm("s_nop 1\n\t"
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, %4\n\t"
      "s_andn2_b64 %2, %4, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2"
      : "+v"(v0), "+v"(v1), "=&s"(sx), "=&s"(sy)
      : "s"(c1)
      : "scc");
  return make64(v0, v1);
}
// y0 * 2^64 - Y for a 32-bit y0 and Y < 2^63: y0*EPS - Y, one borrow correction (the wrapped difference is
// >= 2^64 - 2^63 >= EPS, so it cannot borrow twice).  5 VALU + 1 SALU.
__device__ __forceinline__ u64 fold_hi_asm(u32 y0, u64 Y) {
  u64 M, dm, sx, sy;
  asm("v_mad_u64_u32 %0, %1, %2, -1, 0" : "=v"(M), "=s"(dm) : "v"(y0));
  u32 m0 = (u32)M, m1 = (u32)(M >> 32), q0 = (u32)Y, q1 = (u32)(Y >> 32);
  asm("v_sub_co_u32_e32 %0, vcc, %0, %4\n\ts_nop 1\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %5, vcc\n\ts_nop 1\n\t"   // vcc = bw
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"                // - bw*(2^32-1): lo += bw ...
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"                  // ... hi -= bw & ~carry
      : "+v"(m0), "+v"(m1), "=&s"(sx), "=&s"(sy)
      : "v"(q0), "v"(q1)
      : "vcc", "scc");
  return make64(m0, m1);
}
// x * 2^e for a compile-time e (after unrolling), 0 <= e < 96: the shifts are the compiler's, the folds the ones above.
__device__ __forceinline__ u64 shl_nc_asm(u64 x, int e) {
  if (e == 0) return x;
  if (e < 32) return fold96_asm(x << e, (u32)(x >> (64 - e)));                       // 6 VALU
  if (e < 64) {
    const u64 h = x >> (64 - e);
    return fold128_asm(x << e, (u32)h, (u32)(h >> 32));                                // 10 VALU
  }
  const int f = e - 64;   // x * 2^f = y2 : y1 : y0 (y2 < 2^f);  * 2^64 = y0 * EPS - y1 - y2 * 2^32   (2^96 = -1, 2^128 = -2^32)
  if (f == 0) return fold_hi_asm((u32)x, x >> 32);
  return fold_hi_asm((u32)x << f, x >> (32 - f));                                      // 8 VALU
}
// s = u + v, d = a - b with (a, b) = (u, v) or (v, u): the two carry chains interleaved, each with both corrections.
// 12 VALU + 4 SALU, no compare, no select.
__device__ __forceinline__ void bfly_nc_asm(u64 u, u64 v, bool swap, u64& s, u64& d) {
  const u64 a = swap ? v : u, b = swap ? u : v;
  u32 u0 = (u32)u, u1 = (u32)(u >> 32), v0 = (u32)v, v1 = (u32)(v >> 32);
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u32 s0, s1, d0, d1;
  u64 cA, cB, tA, tB;
  asm("v_add_co_u32_e64 %0, %4, %8, %10\n\t"
      "v_sub_co_u32_e64 %2, %5, %12, %14\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32_e64 %1, %4, %9, %11, %4\n\t"       // cA = carry of u + v
      "v_subb_co_u32_e64 %3, %5, %13, %15, %5\n\t"      // cB = borrow of a - b
      "s_nop 0\n\t"
      "v_subbrev_co_u32_e64 %0, %6, 0, %0, %4\n\t"      // s += cA * EPS: lo -= cA ...
      "v_addc_co_u32_e64 %2, %7, 0, %2, %5\n\t"         // d -= cB * EPS: lo += cB ...
      "s_andn2_b64 %6, %4, %6\n\t"
      "s_andn2_b64 %7, %5, %7\n\t"
      "v_addc_co_u32_e64 %1, %4, 0, %1, %6\n\t"         // ... hi += cA & ~borrow ; cA = second carry
      "v_subbrev_co_u32_e64 %3, %5, 0, %3, %7\n\t"      // ... hi -= cB & ~carry  ; cB = second borrow
      "s_nop 0\n\t"
      "v_subbrev_co_u32_e64 %0, %6, 0, %0, %4\n\t"
      "v_addc_co_u32_e64 %2, %7, 0, %2, %5\n\t"
      "s_andn2_b64 %6, %4, %6\n\t"
      "s_andn2_b64 %7, %5, %7\n\t"
      "v_addc_co_u32_e64 %1, %4, 0, %1, %6\n\t"
      "v_subbrev_co_u32_e64 %3, %5, 0, %3, %7"
      : "=&v"(s0), "=&v"(s1), "=&v"(d0), "=&v"(d1), "=&s"(cA), "=&s"(cB), "=&s"(tA), "=&s"(tB)
      : "v"(u0), "v"(u1), "v"(v0), "v"(v1), "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "scc");
  s = make64(s0, s1);
  d = make64(d0, d1);
}
// ---- single operations (the quotient's gate evaluators) -------------------------------------------------------------------
// a + b for ANY a and b <= p (a canonical value, typically a wire or a constant): the sum wraps at most once (after a
// wrap s = a + b - 2^64 <= b - 1 < p, and s + EPS < 2^64 for s < p).  4 VALU + 1 SALU against the 8 of gl::add.
__device__ __forceinline__ u64 add_c_asm(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 sx, sy;
  asm("v_add_co_u32_e32 %0, vcc, %0, %4\n\ts_nop 1\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %5, vcc\n\ts_nop 1\n\t"
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2"
      : "+v"(a0), "+v"(a1), "=&s"(sx), "=&s"(sy)
      : "v"(b0), "v"(b1)
      : "vcc", "scc");
  return make64(a0, a1);
}
// a - b for ANY a and b <= p: the difference borrows at most once (a second borrow needs d = a - b + 2^64 < EPS, i.e.
// b - a > 2^64 - EPS = p).  4 VALU + 1 SALU against the 6 of gl::sub, and the minuend need not be canonical.
__device__ __forceinline__ u64 sub_c_asm(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 sx, sy;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %4\n\ts_nop 1\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %5, vcc\n\ts_nop 1\n\t"
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"
      : "+v"(a0), "+v"(a1), "=&s"(sx), "=&s"(sy)
      : "v"(b0), "v"(b1)
      : "vcc", "scc");
  return make64(a0, a1);
}
// a - b for ANY a and ANY b: both borrows corrected.  6 VALU + 2 SALU.
__device__ __forceinline__ u64 sub_nc_asm(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 sx, sy;
  asm("v_sub_co_u32_e32 %0, vcc, %0, %4\n\ts_nop 1\n\t"
      "v_subb_co_u32_e32 %1, vcc, %1, %5, vcc\n\ts_nop 1\n\t"
      "v_addc_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2\n\ts_nop 1\n\t"    // %3 = second borrow
      "v_addc_co_u32_e64 %0, %2, 0, %0, %3\n\t"
      "s_andn2_b64 %2, %3, %2\n\t"
      "v_subbrev_co_u32_e64 %1, %3, 0, %1, %2"
      : "+v"(a0), "+v"(a1), "=&s"(sx), "=&s"(sy)
      : "v"(b0), "v"(b1)
      : "vcc", "scc");
  return make64(a0, a1);
}
// a + b for ANY a and ANY b: both wraps corrected.  6 VALU + 2 SALU.
__device__ __forceinline__ u64 add_nc_asm(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 sx, sy;
  asm("v_add_co_u32_e32 %0, vcc, %0, %4\n\ts_nop 1\n\t"
      "v_addc_co_u32_e32 %1, vcc, %1, %5, vcc\n\ts_nop 1\n\t"
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2\n\ts_nop 1\n\t"       // %3 = second carry
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, %3\n\t"
      "s_andn2_b64 %2, %3, %2\n\t"
      "v_addc_co_u32_e64 %1, %3, 0, %1, %2"
      : "+v"(a0), "+v"(a1), "=&s"(sx), "=&s"(sy)
      : "v"(b0), "v"(b1)
      : "vcc", "scc");
  return make64(a0, a1);
}
#endif

// ---- the forms kernels call -------------------------------------------------------------------------------------------
GL_HD void bfly_nc(u64 u, u64 v, bool swap, u64& s, u64& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  bfly_nc_asm(u, v, swap, s, d);
#else
  s = add_nc_c(u, v);
  d = swap ? sub_nc_c(v, u) : sub_nc_c(u, v);
#endif
}
// any + any, any - any
GL_HD u64 add_nc(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return add_nc_asm(a, b);
#else
  return add_nc_c(a, b);
#endif
}
GL_HD u64 sub_nc(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return sub_nc_asm(a, b);
#else
  return sub_nc_c(a, b);
#endif
}
// any + (b <= p), any - (b <= p): one correction.  The caller guarantees the bound on b (a value read from a committed
// matrix, a challenge, a compile-time constant below p).  The host forms are the SAME single-correction definitions and assert
// the bound, so the CPU and sanitizer builds exercise the contract the gfx950 sequences rely on (a caller that broke it would
// otherwise go wrong on the device only); tools/exp/lazy_contract_check.patch adds the same checks -- and the wave-uniformity
// of mad_nc_s's second factor -- to the device paths for a debug run of the GPU suite.
GL_HD u64 add_c(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return add_c_asm(a, b);
#else
  assert(b <= P && "gl::add_c: second operand above p");
  return add_c_c(a, b);
#endif
}
GL_HD u64 sub_c(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return sub_c_asm(a, b);
#else
  assert(b <= P && "gl::sub_c: second operand above p");
  return sub_c_c(a, b);
#endif
}
// a - k (mod 2^64, NOT mod p) for a canonical a and a small k: a factor of a vanishing product a (a - 1) ... (a - m), m >= k.
// The difference wraps only for a < k, and then the product holds the factor (a - a) = 0 exactly, so the product is
// 0 = the true value whatever this factor reads -- the modular correction of gl::sub (5 instructions) is never needed.
GL_HD u64 dec_wrap(u64 a, u64 k) { return a - k; }
GL_HD u64 shl_nc(u64 x, int e) {
#if defined(__HIP_DEVICE_COMPILE__)
  return shl_nc_asm(x, e);
#else
  return shl_nc_c(x, e);
#endif
}

// x * y in F_p[X] / (X^2 - 7), any-u64 components in and out: five multiplications, two of them fused multiply-adds,
// no canonical step (gl::mul(E2, E2) pays five canonical products and two canonical additions: 106 VALU against 74)
GL_HD E2 e2_mul_nc(E2 x, E2 y) {
  const u64 a = mad_nc_s(mul_nc(x.b, y.b), EXT_W, mul_nc(x.a, y.a));
  const u64 b = mad_nc(x.a, y.b, mul_nc(x.b, y.a));
  return E2{a, b};
}
GL_HD E2 e2_canon(E2 x) { return E2{canon(x.a), canon(x.b)}; }

}  // namespace gl
