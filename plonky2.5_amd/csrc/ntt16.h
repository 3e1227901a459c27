// Radix-16 DIF transform on compile-time twiddles, in lazy (non-canonical) arithmetic (gl_lazy.h).
#pragma once
#include "gl_lazy.h"

namespace p25 {

// ---- radix-16 groups on compile-time twiddles --------------------------------------------------------------------
// In Goldilocks 2 is a 192nd root of unity (2^96 = -1), so every 16th root of unity is a signed power of two: the
// library's w_16 (root_of_unity(4)) is 2^156 = -2^60 and its inverse 2^36.  A 16-point DIF network therefore needs
// no table and no general multiplication: (u - v) * w_16^J is a shift by a constant followed by a fold of the bits
// above 2^64 (gl::shl_nc: 6 / 10 / 8 VALU for exponents below 32 / below 64 / below 96), and the sign is absorbed by
// swapping the operands of the subtraction.  Round 5: every value inside the network is ANY u64 congruent to the
// element (the canonical form cost 1,043 VALU per 16 points, this one 530: 32 sum/difference pairs of 12 and the
// seventeen non-trivial shifts 146).
//
// Pure 16-point DIF transform of x[0..16) in registers: natural in, bit-reversed out (x[k'] holds frequency rev4(k')).
template <bool INV>
GL_HD void dft16(u64 (&x)[16]) {
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int hk = 8 >> m;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (k & hk) continue;
      const int J = (k & (hk - 1)) << m;
      const int e = ((INV ? 36 : 156) * J) % 192;   // w_16^J = 2^e, and 2^96 = -1
      u64 s, d;
      gl::bfly_nc(x[k], x[k + hk], e >= 96, s, d);
      x[k] = s;
      x[k + hk] = gl::shl_nc(d, e >= 96 ? e - 96 : e);
    }
  }
}
}  // namespace p25
