// Radix-16 DIF transform on compile-time twiddles (shared by kernels_ntt.hip and its host-side unit test).
#pragma once
#include "gl.h"

namespace p25 {

// ---- radix-16 groups on compile-time twiddles --------------------------------------------------------------------
// In Goldilocks 2 is a 192nd root of unity (2^96 = -1), so every 16th root of unity is a signed power of two: the
// library's w_16 (root_of_unity(4)) is 2^156 = -2^60 and its inverse 2^36.  A 16-point DIF network therefore needs
// no table and no general multiplication: (u - v) * w_16^J is a shift by a constant followed by the 128-bit
// reduction, and the sign is absorbed by swapping the operands of the subtraction.  Four of the seventeen
// non-trivial factors have an exponent >= 64 and go through the ordinary multiply with an immediate constant.
GL_HD u64 mul_pow2(u64 x, int e) {  // x * 2^e, 0 <= e < 96, canonical in/out; e is a constant after unrolling
  if (e == 0) return x;
  if (e < 64) return gl::canon(gl::reduce128(x << e, x >> (64 - e)));
  return gl::mul(x, (u64)0xFFFFFFFFull << (e - 64));  // 2^64 = 2^32 - 1 (mod p)
}
template <bool INV>
GL_HD u64 diff_times_w16(u64 u, u64 v, int J) {  // (u - v) * w_16^J (inverse root if INV)
  const int e = ((INV ? 36 : 156) * J) % 192;
  return e >= 96 ? mul_pow2(gl::sub(v, u), e - 96) : mul_pow2(gl::sub(u, v), e);
}
// Pure 16-point DIF transform of x[0..16) in registers: natural in, bit-reversed out (x[k'] holds frequency rev4(k')).
template <bool INV>
GL_HD void dft16(u64 (&x)[16]) {
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int hk = 8 >> m;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k & hk) continue;
      const u64 u = x[k], v = x[k + hk];
      x[k] = gl::add(u, v);
      x[k + hk] = diff_times_w16<INV>(u, v, (k & (hk - 1)) << m);
    }
  }
}
}  // namespace p25
