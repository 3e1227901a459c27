// ORACLE -- test infrastructure only.  Goldilocks arithmetic on eight canonical field elements at a time (one per 64-bit
// lane of an AVX-512 register) for the TUNED cpu_baseline leg (ref_hash_x8.cpp, ref_quotient_x8.cpp).  Every function takes
// and returns canonical values (< p), so results are bit-identical to ref_field.h's scalar rf_add / rf_sub / rf_mul.
// Include only from code compiled for AVX-512 F + DQ (function target attributes or a `#pragma GCC target`).
#pragma once
#include <immintrin.h>
#include "ref_field.h"

#define X8 __attribute__((target("avx512f,avx512dq"), always_inline)) static inline
typedef __m512i V;

X8 V v_eps() { return _mm512_set1_epi64(0xFFFFFFFFLL); }
X8 V v_p() { return _mm512_set1_epi64((long long)0xFFFFFFFF00000001ULL); }

// canonical a, b -> canonical a + b
X8 V v_add(V a, V b) {
  V s = _mm512_add_epi64(a, b);
  __mmask8 c = _mm512_cmplt_epu64_mask(s, a);
  s = _mm512_mask_add_epi64(s, c, s, v_eps());          // wrapped 2^64: + (2^32 - 1) = - p (mod 2^64)
  __mmask8 ge = _mm512_cmpge_epu64_mask(s, v_p());
  return _mm512_mask_sub_epi64(s, ge, s, v_p());
}
// (hi, lo) = a 128-bit value with hi, lo < 2^64 -> canonical residue: 2^64 = 2^32 - 1, 2^96 = -1 (mod p)
X8 V v_reduce128(V hi, V lo) {
  const V eps = v_eps();
  V hi_hi = _mm512_srli_epi64(hi, 32), hi_lo = _mm512_and_si512(hi, eps);
  V t = _mm512_sub_epi64(lo, hi_hi);
  __mmask8 b = _mm512_cmplt_epu64_mask(lo, hi_hi);
  t = _mm512_mask_sub_epi64(t, b, t, eps);
  V m = _mm512_mul_epu32(hi_lo, eps);                   // hi_lo * (2^32 - 1)
  V r = _mm512_add_epi64(t, m);
  __mmask8 c = _mm512_cmplt_epu64_mask(r, m);
  r = _mm512_mask_add_epi64(r, c, r, eps);
  __mmask8 ge = _mm512_cmpge_epu64_mask(r, v_p());
  return _mm512_mask_sub_epi64(r, ge, r, v_p());
}
X8 V v_mul(V a, V b) {
  const V eps = v_eps();
  V ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
  V ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
  V t0 = _mm512_add_epi64(hl, _mm512_srli_epi64(ll, 32));
  V t1 = _mm512_add_epi64(lh, _mm512_and_si512(t0, eps));
  V hi = _mm512_add_epi64(_mm512_add_epi64(hh, _mm512_srli_epi64(t0, 32)), _mm512_srli_epi64(t1, 32));
  V lo = _mm512_or_si512(_mm512_and_si512(ll, eps), _mm512_slli_epi64(t1, 32));
  return v_reduce128(hi, lo);
}
// canonical a, b -> canonical a - b
X8 V v_sub(V a, V b) {
  V d = _mm512_sub_epi64(a, b);
  __mmask8 bw = _mm512_cmplt_epu64_mask(a, b);
  return _mm512_mask_sub_epi64(d, bw, d, v_eps());      // wrapped a - b + 2^64: - (2^32 - 1) = + p (mod 2^64)
}
