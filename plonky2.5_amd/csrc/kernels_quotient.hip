// Quotient-polynomial evaluation on the LDE coset: one lane per coset point.
//
// Replaces upstream plonky2 @ 3de92d9 prover.rs `compute_quotient_polys` +
// vanishing_poly.rs `eval_vanishing_poly_base_batch` (reached from
// /root/reference/src/p3/mod.rs:260) and the gate callbacks it invokes -- the reference's own
//   Poseidon2Gate::eval_unfiltered_base_one          src/common/poseidon2/poseidon2_gate.rs:233-310
//   U32ArithmeticGate::eval_unfiltered_base_packed   src/common/u32/gates/arithmetic_u32.rs:303-366
//   U32InterleaveGate::eval_unfiltered_base_packed   src/common/u32/gates/interleave_u32.rs:250-287
//   UninterleaveToU32Gate::eval_unfiltered_base_packed  src/common/u32/gates/uninterleave_to_u32.rs:285-335
// plus upstream's Noop / Constant / PublicInput / BaseSum<2> / Arithmetic / MulExtension /
// Exponentiation gates.
//
// Data layout: every committed matrix is column-major at bit-reversed leaf positions, so lane p of a
// wave reads word p of each column: 242 fully coalesced column reads per point, no transposes.
// Instead of materialising the <=134 per-gate constraint slots and reducing them afterwards
// (upstream's reduce_with_powers_multi), each gate folds its constraints into the two alpha-power
// sums as they are produced (alpha^j from an LDS table), and the gate's filter multiplies the
// folded sums once:   sum_j alpha^j sum_g f_g c_{g,j} = sum_g f_g sum_j alpha^j c_{g,j}.
#include <stdexcept>
#include <stdlib.h>
#include "builder.h"
#include "kernels.h"
#include "gl_lazy.h"
#include "poseidon.h"
#include "poseidon2.h"
#include "coop.h"
#include "prover_kernels.h"

namespace p25 {

namespace {

// alpha^j for both challenges as 22/22/20-bit limbs (8 x u32 per j, LDS): a constraint value c (any
// u64, as two 32-bit halves) is folded into the alpha-sums with 12 carry-free v_mad_u64_u32 --
// products are < 2^54, so 512 terms fit a 64-bit accumulator -- instead of two 64x64->128 multiplies
// with 128-bit accumulation and overflow tracking (~40 instructions).  Reduced mod p once per gate.
struct AlphaLimbs {
  u32 l[8];  // [challenge 0: l0 l1 l2 pad | challenge 1: l0 l1 l2 pad]
};
constexpr int MAX_TERMS_PER_FOLD = 512;

// A committed matrix (column-major, leaf order) seen from ONE lane: column c at this lane's point.  Read through a
// buffer descriptor per column -- base (a kernel argument) + c * big on the scalar unit, the lane's share a 32-bit BYTE
// offset -- so a column read is `buffer_load_dwordx2 v, v_off, s[desc], 0 offen`: its address lives in SGPRs.  The
// per-lane pointer form (`(base + p)[c * big]`) cost a 64-bit VALU add and a VGPR pair per read, and the compiler hoisted
// those column addresses out of the loop over the gates: hundreds of live registers, i.e. the kernel's spills.
struct LaneCols {
  const u64* base;
  u32 boff;  // 8 * point index (the LDE has at most 2^22 points)
  size_t big;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  __device__ __forceinline__ u64 col(size_t c) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t r =
        __builtin_amdgcn_make_buffer_rsrc((void*)const_cast<u64*>(base + c * big), 0, (int)(big * 8), 0x00020000);
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)boff, 0, 0);
    return ((u64)v.y << 32) | v.x;
#else
    return 0;
#endif
  }
};
struct Ctx {
  LaneCols wires;  // wires.col(c) = wire column c at this point
  size_t big;
  const AlphaLimbs* apl;  // LDS
  u64 acc[2][2][3];       // [challenge][half of c][limb of alpha]
  __device__ __forceinline__ u64 w(int col) const { return wires.col((size_t)col); }
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int k = 0; k < 3; k++) acc[c][h][k] = 0;
  }
  __device__ __forceinline__ static void mad(u64& a, u32 x, u32 y) {
    u64 dm;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a), "=s"(dm) : "v"(x), "v"(y));
  }
  // c: any u64 congruent to the constraint value
  __device__ __forceinline__ void at(int j, u64 c) {
    const AlphaLimbs al = apl[j];
    const u32 c0 = (u32)c, c1 = (u32)(c >> 32);
#pragma unroll
    for (int ch = 0; ch < 2; ch++)
#pragma unroll
      for (int k = 0; k < 3; k++) {
        mad(acc[ch][0][k], c0, al.l[4 * ch + k]);
        mad(acc[ch][1][k], c1, al.l[4 * ch + k]);
      }
  }
  // sum_{h,k} acc[h][k] * 2^(32h + 22k) mod p
  // (any u64 congruent to the sum: every consumer multiplies it)
  __device__ __forceinline__ u64 fold(int ch) const {
    u64 r = acc[ch][0][0];
    r = gl::mad_nc_s(acc[ch][0][1], (u64)1 << 22, r);
    r = gl::mad_nc_s(acc[ch][0][2], (u64)1 << 44, r);
    r = gl::mad_nc_s(acc[ch][1][0], (u64)1 << 32, r);
    r = gl::mad_nc_s(acc[ch][1][1], (u64)1 << 54, r);
    // 2^76 = 2^64 * 2^12 = (2^32 - 1) * 2^12 (mod p)
    r = gl::mad_nc_s(acc[ch][1][2], (u64)0xFFFFFFFFull << 12, r);
    return r;
  }
  __device__ __forceinline__ u64 acc0() const { return fold(0); }
  __device__ __forceinline__ u64 acc1() const { return fold(1); }
};

// Lazy arithmetic (gl_lazy.h) throughout the evaluators: a constraint value goes into the alpha fold (`at`) as ANY u64
// congruent to it, so nothing here needs a canonical result; what the single-correction forms gl::add_c / gl::sub_c need is
// a second operand <= p, and every wire, constant column, sigma, challenge and public-input hash word read from memory is
// canonical.  `bool01(b)` = b (b - 1) and the base-4 range products use gl::dec_wrap (see there).
__device__ __forceinline__ u64 bool01(u64 b) { return gl::mul_nc(b, gl::dec_wrap(b, 1)); }
__device__ void gate_constant(Ctx& cx, u64 k0, u64 k1) {
  cx.at(0, gl::sub_c(k0, cx.w(0)));
  cx.at(1, gl::sub_c(k1, cx.w(1)));
}
__device__ void gate_public_input(Ctx& cx, const u64* __restrict__ pih) {
  // wire_i - public_inputs_hash_i (upstream gates/public_input.rs); the hash of the empty list is [0, 0, 0, 0]
  for (int i = 0; i < 4; i++) cx.at(i, gl::sub_c(cx.w(i), pih[i]));
}
__device__ void gate_base_sum(Ctx& cx) {
  // sum_i limb_i 2^i, i < 63: carry-free groups of eight limbs (cf. SmallLin below), most significant group
  // first, joined by sum = sum * 2^8 + group
  u64 sum = 0;
  for (int q = (BASE_SUM_LIMBS + 7) / 8 - 1; q >= 0; q--) {
    u64 al = 0, ah = 0, dm;
    const int n = BASE_SUM_LIMBS - 8 * q < 8 ? BASE_SUM_LIMBS - 8 * q : 8;
    for (int t = 0; t < n; t++) {
      const int i = 8 * q + t;
      u64 l = cx.w(1 + i);
      const u32 c = 1u << t;
      asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(al), "=s"(dm) : "v"((u32)l), "s"(c));
      asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(ah), "=s"(dm) : "v"((u32)(l >> 32)), "s"(c));
      cx.at(1 + i, bool01(l));
    }
    sum = gl::mad_nc(sum, (u64)1 << 8, coop::reduce_row(al, ah));
  }
  cx.at(0, gl::sub_c(sum, cx.w(0)));
}
__device__ void gate_arithmetic(Ctx& cx, u64 k0, u64 k1) {
  for (int i = 0; i < 20; i++) {
    u64 m0 = cx.w(4 * i), m1 = cx.w(4 * i + 1), ad = cx.w(4 * i + 2), o = cx.w(4 * i + 3);
    u64 comp = gl::mad_nc(gl::mul_nc(m0, m1), k0, gl::mul_nc(ad, k1));
    cx.at(i, gl::sub_nc(o, comp));
  }
}
// (a * b) * k in F_p^2 = F_p[X] / (X^2 - 7), any-u64 components: 8 multiplications, no canonical step
__device__ __forceinline__ gl::E2 e2_mul_scaled(gl::E2 x, gl::E2 y, u64 k) {
  const u64 ra = gl::mad_nc_s(gl::mul_nc(x.b, y.b), gl::EXT_W, gl::mul_nc(x.a, y.a));
  const u64 rb = gl::mad_nc(x.a, y.b, gl::mul_nc(x.b, y.a));
  return gl::E2{gl::mul_nc(ra, k), gl::mul_nc(rb, k)};
}
__device__ void gate_mul_ext(Ctx& cx, u64 k0) {
  for (int i = 0; i < 13; i++) {
    gl::E2 a{cx.w(6 * i), cx.w(6 * i + 1)}, b{cx.w(6 * i + 2), cx.w(6 * i + 3)};
    gl::E2 p = e2_mul_scaled(a, b, k0);
    cx.at(2 * i, gl::sub_nc(cx.w(6 * i + 4), p.a));
    cx.at(2 * i + 1, gl::sub_nc(cx.w(6 * i + 5), p.b));
  }
}
// Every evaluator below reads its wires ONE ITERATION AHEAD (`nb`, `ni`, `nxt` ...): left to itself the compiler emits
// each column load right where its value is used, followed by s_waitcnt vmcnt(0) -- one exposed memory round trip per
// wire and ~1,000 of them per wave (profiles/r03_pipeline_model_experiments.txt); with the next wire's load issued before
// the current wire's 40-90 instructions of arithmetic, two loads per wave are in flight and the trip hides behind them.
__device__ void gate_exponentiation(Ctx& cx) {
  const u64 base = cx.w(0);
  u64 prev_inter = 1;
  u64 nb = cx.w(1 + (EXP_POWER_BITS - 1)), ni = cx.w(2 + EXP_POWER_BITS);
  for (int i = 0; i < EXP_POWER_BITS; i++) {
    u64 prev = i == 0 ? 1 : gl::mul_nc(prev_inter, prev_inter);  // intermediate values may stay non-canonical
    const u64 bit = nb, inter = ni;
    if (i + 1 < EXP_POWER_BITS) {
      nb = cx.w(1 + (EXP_POWER_BITS - 2 - i));
      ni = cx.w(3 + EXP_POWER_BITS + i);
    }
    u64 sel = gl::mad_nc(bit, base, gl::sub_c(1, bit));            // bit * base + (1 - bit)
    cx.at(i, gl::sub_c(gl::mul_nc(prev, sel), inter));
    prev_inter = inter;
  }
  cx.at(EXP_POWER_BITS, gl::sub_c(cx.w(1 + EXP_POWER_BITS), prev_inter));
}
// sum_i v_i * c_i for small constants c_i (wave-uniform, < 2^15) and 64-bit field values v_i: two carry-free
// multiply-adds per term on the 32-bit halves, one 5-instruction reduction at the end -- instead of the
// Horner chains x = 2x + b (two modular additions, 12 instructions, per step).  At most 16 terms per group:
// products < 2^47, sums < 2^51.
struct SmallLin {
  u64 al = 0, ah = 0;
  __device__ __forceinline__ void add(u64 v, u32 c_uniform) {
    u64 dm;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(al), "=s"(dm) : "v"((u32)v), "s"(c_uniform));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(ah), "=s"(dm) : "v"((u32)(v >> 32)), "s"(c_uniform));
  }
  __device__ __forceinline__ u64 value() const { return coop::reduce_row(al, ah); }  // any u64 representative
};
// ((g[0] * 2^s + g[1]) * 2^s + g[2]) * 2^s + g[3]
__device__ __forceinline__ u64 join4(const u64 g[4], u32 shift) {
  const u64 m = (u64)1 << shift;
  return gl::mad_nc(gl::mad_nc(gl::mad_nc(g[0], m, g[1]), m, g[2]), m, g[3]);
}

__device__ void gate_u32_arithmetic(Ctx& cx) {
  for (int i = 0; i < 3; i++) {
    const int cb = 36 * i;
    u64 m0 = cx.w(6 * i), m1 = cx.w(6 * i + 1), ad = cx.w(6 * i + 2);
    u64 lo = cx.w(6 * i + 3), hi = cx.w(6 * i + 4), inv = cx.w(6 * i + 5);
    u64 computed = gl::mad_nc(m0, m1, ad);
    u64 diff = gl::sub_c(0xFFFFFFFFull, hi);
    u64 hi_not_max = gl::sub_c(gl::mul_nc(inv, diff), 1);
    cx.at(cb, gl::mul_nc(hi_not_max, lo));
    u64 combined = gl::mad_nc_s(hi, (u64)1 << 32, lo);
    cx.at(cb + 1, gl::sub_nc(combined, computed));
    // 32 base-4 limbs, most significant first: limbs 31..16 make the high word, 15..0 the low word
    u64 part[4];
    u64 nxt = cx.w(18 + 32 * i + 31);
#pragma unroll
    for (int q = 0; q < 4; q++) {  // limbs 31-8q .. 24-8q
      SmallLin acc;
      for (int t = 0; t < 8; t++) {
        const int j = 31 - 8 * q - t;
        const u64 l = nxt;
        if (j > 0) nxt = cx.w(18 + 32 * i + j - 1);
        u64 pr = gl::mul_nc(gl::mul_nc(l, gl::dec_wrap(l, 1)), gl::mul_nc(gl::dec_wrap(l, 2), gl::dec_wrap(l, 3)));
        cx.at(cb + 2 + (31 - j), pr);
        acc.add(l, 1u << (2 * (7 - t)));
      }
      part[q] = acc.value();
    }
    const u64 ch = gl::mad_nc(part[0], (u64)1 << 16, part[1]), cl = gl::mad_nc(part[2], (u64)1 << 16, part[3]);
    cx.at(cb + 34, gl::sub_c(cl, lo));
    cx.at(cb + 35, gl::sub_c(ch, hi));
  }
}
__device__ void gate_u32_interleave(Ctx& cx) {
  for (int i = 0; i < 3; i++) {
    const int cb = 34 * i;
    u64 xq[4], xiq[4];  // bits 8q .. 8q+7, most significant first
    u64 nxt = cx.w(6 + 32 * i);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      SmallLin ax, axi;
      for (int t = 0; t < 8; t++) {
        const int b = 8 * q + t;
        const u64 bit = nxt;
        if (b < 31) nxt = cx.w(6 + 32 * i + b + 1);
        ax.add(bit, 1u << (7 - t));
        axi.add(bit, 1u << (2 * (7 - t)));
        cx.at(cb + 2 + b, bool01(bit));
      }
      xq[q] = ax.value();
      xiq[q] = axi.value();
    }
    cx.at(cb, gl::sub_c(join4(xq, 8), cx.w(2 * i)));
    cx.at(cb + 1, gl::sub_c(join4(xiq, 16), cx.w(2 * i + 1)));
  }
}
__device__ void gate_u32_uninterleave(Ctx& cx) {
  for (int i = 0; i < 2; i++) {
    const int cb = 67 * i;
    u64 xq[4], evq[4], odq[4];  // bit pairs 8q .. 8q+7, most significant first
    u64 ne = cx.w(6 + 64 * i), no = cx.w(6 + 64 * i + 1);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      SmallLin ax, aev, aod;
      for (int t = 0; t < 8; t++) {
        const int j = 8 * q + t;
        const u64 be = ne, bo = no;
        if (j < 31) {
          ne = cx.w(6 + 64 * i + 2 * j + 2);
          no = cx.w(6 + 64 * i + 2 * j + 3);
        }
        ax.add(be, 2u << (2 * (7 - t)));
        ax.add(bo, 1u << (2 * (7 - t)));
        aev.add(be, 1u << (7 - t));
        aod.add(bo, 1u << (7 - t));
        cx.at(cb + 3 + 2 * j, bool01(be));
        cx.at(cb + 3 + 2 * j + 1, bool01(bo));
      }
      xq[q] = ax.value();
      evq[q] = aev.value();
      odq[q] = aod.value();
    }
    cx.at(cb, gl::sub_c(join4(xq, 16), cx.w(3 * i)));
    cx.at(cb + 1, gl::sub_c(join4(evq, 8), cx.w(3 * i + 1)));
    cx.at(cb + 2, gl::sub_c(join4(odq, 8), cx.w(3 * i + 2)));
  }
}
// ---- Poseidon2's linear layers for the gate evaluator, in lazy arithmetic (any u64 in, any u64 out) -----------------------
// The external layer (poseidon2.rs:126-147: M4 on each block of four, then the column sums) is the 12 x 12 matrix
// circ(2 M4, M4, M4), M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]: entries <= 14, row sums 64.  As 62 modular additions it
// costs ~500 VALU; on the 32-bit halves it is 24 carry-free v_mad_u64_u32 with inline-constant entries per output word and
// the 5-instruction row reduction of poseidon_p3r.h -- 348 -- and the constants the NEXT round adds ride in as the addend of
// each row's first multiply-add, read from SGPRs, exactly as in the Poseidon MDS layer (poseidon.h: mds_rc).
namespace p2lazy {
#if defined(__HIP_DEVICE_COMPILE__)
constexpr int ext_coef(int i, int j) {
  constexpr int M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
  return M4[i & 3][j & 3] * ((i >> 2) == (j >> 2) ? 2 : 1);
}
// rows of zero-extended (lo, hi) halves: 0..7 the full rounds' constants, 8 = (RC_MID[0], 0, ..., 0) (the layer that
// leads into the partial rounds), 9 = zeros (the last layer)
struct RcSplit {
  u64 v[10 * 12 * 2];
};
constexpr RcSplit make_rc_split() {
  RcSplit t{};
  for (int i = 0; i < 96; i++) {
    t.v[2 * i] = poseidon2::P2_RC[i] & 0xFFFFFFFFull;
    t.v[2 * i + 1] = poseidon2::P2_RC[i] >> 32;
  }
  t.v[2 * 96] = poseidon2::P2_RC_MID[0] & 0xFFFFFFFFull;
  t.v[2 * 96 + 1] = poseidon2::P2_RC_MID[0] >> 32;
  return t;
}
static constexpr RcSplit RC_SPLIT = make_rc_split();
typedef const u64 __attribute__((address_space(4))) * rc_ptr;   // constant address space: scalar loads

template <int I, int J>
__device__ __forceinline__ void ext_terms(u64& al, u64& ah, const u32* lo, const u32* hi) {
  if constexpr (J < 12) {
    poseidon::mad_k<ext_coef(I, J)>(al, lo[J]);
    poseidon::mad_k<ext_coef(I, J)>(ah, hi[J]);
    ext_terms<I, J + 1>(al, ah, lo, hi);
  }
}
template <int I>
__device__ __forceinline__ void ext_rows(u64* s, const u32* lo, const u32* hi, rc_ptr k) {
  if constexpr (I < 12) {
    u64 al, ah, dm;
    asm("v_mad_u64_u32 %0, %1, %2, %4, %3" : "=v"(al), "=s"(dm) : "v"(lo[0]), "s"(k[2 * I]), "n"(ext_coef(I, 0)));
    asm("v_mad_u64_u32 %0, %1, %2, %4, %3" : "=v"(ah), "=s"(dm) : "v"(hi[0]), "s"(k[2 * I + 1]), "n"(ext_coef(I, 0)));
    ext_terms<I, 1>(al, ah, lo, hi);
    s[I] = poseidon::p3r::reduce_row(al, ah);   // al, ah < 2^32 * 65
    ext_rows<I + 1>(s, lo, hi, k);
  }
}
// s <- E s + (constants row `row` of RC_SPLIT)
__device__ __forceinline__ void external(u64 s[12], int row) {
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  ext_rows<0>(s, lo, hi, (rc_ptr)RC_SPLIT.v + 24 * row);
}
// s_i <- s_i (d_i - 1) + sum_j s_j (poseidon2.rs:163-182): the sum carry-free on the halves, one fused multiply-add per
// word with the diagonal entry read from SGPRs; no canonical step
template <int I>
__device__ __forceinline__ void int_rows(u64* s, u64 sum) {
  if constexpr (I < 12) {
    s[I] = gl::mad_nc_s(s[I], poseidon2::P2_MAT_DIAG_M_1[I] - 1, sum);
    int_rows<I + 1>(s, sum);
  }
}
__device__ __forceinline__ void internal(u64 s[12]) {
  u64 al = 0, ah = 0;
#pragma unroll
  for (int i = 0; i < 12; i++) {
    poseidon::mad_k<1>(al, (u32)s[i]);
    poseidon::mad_k<1>(ah, (u32)(s[i] >> 32));
  }
  int_rows<0>(s, poseidon::p3r::reduce_row(al, ah));
}
__device__ __forceinline__ u64 sbox(u64 x) {   // x^7, any u64 in and out
  const u64 x2 = gl::mul_nc(x, x), x4 = gl::mul_nc(x2, x2), x3 = gl::mul_nc(x, x2);
  return gl::mul_nc(x3, x4);
}
#else   // the host pass only parses the kernels
__device__ __forceinline__ void external(u64*, int) {}
__device__ __forceinline__ void internal(u64*) {}
__device__ __forceinline__ u64 sbox(u64 x) { return x; }
#endif
}  // namespace p2lazy

// poseidon2_gate.rs:233-310
__device__ __forceinline__ void gate_poseidon2(Ctx& cx) {
  using namespace poseidon2;
  int nc = 0;
  u64 swap = cx.w(24);
  cx.at(nc++, bool01(swap));
  u64 st[12];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u64 lhs = cx.w(i), rhs = cx.w(i + 4), delta = cx.w(25 + i);
    cx.at(nc++, gl::sub_c(gl::mul_nc(swap, gl::sub_c(rhs, lhs)), delta));
    st[i] = gl::add_c(lhs, delta);
    st[i + 4] = gl::sub_c(rhs, delta);
  }
#pragma unroll
  for (int i = 8; i < 12; i++) st[i] = cx.w(i);
  p2lazy::external(st, 0);              // ... + the constants of round 0
  for (int r = 0; r < ROUND_F_BEGIN; r++) {
    if (r != 0) {
      u64 nxt = cx.w(29 + 12 * (r - 1));
#pragma unroll
      for (int i = 0; i < 12; i++) {
        const u64 sb = nxt;
        if (i < 11) nxt = cx.w(29 + 12 * (r - 1) + i + 1);
        cx.at(nc++, gl::sub_c(st[i], sb));
        st[i] = sb;
      }
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = p2lazy::sbox(st[i]);
    p2lazy::external(st, r + 1 < ROUND_F_BEGIN ? r + 1 : 8);   // ... + round r + 1's constants / RC_MID[0] on word 0
  }
  u64 nsb = cx.w(65);
  for (int r = 0; r < ROUND_P; r++) {
    const u64 sb = nsb;
    if (r + 1 < ROUND_P) nsb = cx.w(66 + r);
    cx.at(nc++, gl::sub_c(st[0], sb));
    st[0] = p2lazy::sbox(sb);
    p2lazy::internal(st);
    if (r + 1 < ROUND_P) st[0] = gl::add_c(st[0], P2_RC_MID[r + 1]);
  }
  for (int r = ROUND_F_BEGIN; r < ROUND_F_END; r++) {
    if (r == ROUND_F_BEGIN) {   // the internal layer carries no constants: round 4's are added here
#pragma unroll
      for (int i = 0; i < 12; i++) st[i] = gl::add_c(st[i], P2_RC[12 * ROUND_F_BEGIN + i]);
    }
    u64 nxt = cx.w(87 + 12 * (r - ROUND_F_BEGIN));
#pragma unroll
    for (int i = 0; i < 12; i++) {
      const u64 sb = nxt;
      if (i < 11) nxt = cx.w(87 + 12 * (r - ROUND_F_BEGIN) + i + 1);
      cx.at(nc++, gl::sub_c(st[i], sb));
      st[i] = sb;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = p2lazy::sbox(st[i]);
    p2lazy::external(st, r + 1 < ROUND_F_END ? r + 1 : 9);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) cx.at(nc++, gl::sub_c(st[i], cx.w(12 + i)));
}

// upstream gates/arithmetic_extension.rs: out - (c0 * m0 * m1 + c1 * addend) in F_p^2, 10 ops of 8 wires
__device__ void gate_arith_ext(Ctx& cx, u64 k0, u64 k1) {
  u64 nx[8];
#pragma unroll
  for (int k = 0; k < 8; k++) nx[k] = cx.w(k);
  for (int i = 0; i < 10; i++) {
    u64 w[8];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = nx[k];
    if (i + 1 < 10) {
#pragma unroll
      for (int k = 0; k < 8; k++) nx[k] = cx.w(8 * (i + 1) + k);
    }
    gl::E2 a{w[0], w[1]}, b{w[2], w[3]}, ad{w[4], w[5]};
    gl::E2 r = gl::add(gl::mul(gl::mul(a, b), k0), gl::mul(ad, k1));
    cx.at(2 * i, gl::sub(w[6], r.a));
    cx.at(2 * i + 1, gl::sub(w[7], r.b));
  }
}
// upstream gates/random_access.rs eval_unfiltered_base_one: per copy the index bits are boolean and recompose the
// index, and folding the list by them (x + b (y - x) on adjacent pairs, least significant bit first) leaves the
// claimed element; then the extra constant wires equal the row's constants.
__device__ void gate_random_access(Ctx& cx, u64 k0, u64 k1) {
  int nc = 0;
  for (int copy = 0; copy < RA_COPIES; copy++) {
    const int base = (2 + RA_VEC) * copy;
    u64 bits[RA_BITS];
    for (int i = 0; i < RA_BITS; i++) {
      bits[i] = cx.w(RA_ROUTED + RA_BITS * copy + i);
      cx.at(nc++, gl::mul_nc(bits[i], gl::sub(bits[i], 1)));
    }
    u64 idx = 0;
    for (int i = RA_BITS - 1; i >= 0; i--) idx = gl::add(gl::add(idx, idx), bits[i]);
    cx.at(nc++, gl::sub(idx, cx.w(base)));
    u64 items[RA_VEC];
    for (int i = 0; i < RA_VEC; i++) items[i] = cx.w(base + 2 + i);
    int len = RA_VEC;
    for (int bi = 0; bi < RA_BITS; bi++) {
      len >>= 1;
      for (int i = 0; i < len; i++)
        items[i] = gl::add(items[2 * i], gl::mul(bits[bi], gl::sub(items[2 * i + 1], items[2 * i])));
    }
    cx.at(nc++, gl::sub(items[0], cx.w(base + 1)));
  }
  cx.at(nc++, gl::sub(k0, cx.w((2 + RA_VEC) * RA_COPIES)));
  cx.at(nc++, gl::sub(k1, cx.w((2 + RA_VEC) * RA_COPIES + 1)));
}
// upstream gates/reducing.rs / reducing_extension.rs: acc_i = acc_{i-1} * alpha + coeff_i in the extension field
// (old_acc first, the last accumulator is the output pair at wires 0, 1); two base-field constraints per step.
template <bool EXT>
__device__ void gate_reducing(Ctx& cx) {
  constexpr int NCO = EXT ? REDX_COEFFS : RED_COEFFS, CW = EXT ? 2 : 1, START_ACCS = 6 + NCO * CW;
  const gl::E2 alpha{cx.w(2), cx.w(3)};
  gl::E2 acc{cx.w(4), cx.w(5)};
  auto acc_wire = [](int i) { return i == NCO - 1 ? 0 : START_ACCS + 2 * i; };
  u64 na = cx.w(acc_wire(0)), nb = cx.w(acc_wire(0) + 1), nc0 = cx.w(6), nc1 = EXT ? cx.w(7) : 0;
  for (int i = 0; i < NCO; i++) {
    const gl::E2 nxt{na, nb};
    const u64 c0 = nc0, c1 = nc1;
    if (i + 1 < NCO) {
      na = cx.w(acc_wire(i + 1));
      nb = cx.w(acc_wire(i + 1) + 1);
      nc0 = cx.w(6 + CW * (i + 1));
      if (EXT) nc1 = cx.w(6 + CW * (i + 1) + 1);
    }
    gl::E2 t = gl::mul(acc, alpha);
    t.a = gl::add(t.a, c0);
    if (EXT) t.b = gl::add(t.b, c1);
    cx.at(2 * i, gl::sub(t.a, nxt.a));
    cx.at(2 * i + 1, gl::sub(t.b, nxt.b));
    acc = nxt;
  }
}
// upstream gates/poseidon_mds.rs: output_r - (sum_i circ[i] input_{(i + r) mod 12} + diag[r] input_r), per component
__device__ void gate_poseidon_mds(Ctx& cx) {
  for (int d = 0; d < 2; d++) {
    u64 in[12];
#pragma unroll
    for (int i = 0; i < 12; i++) in[i] = cx.w(2 * i + d);
    u64 nxt = cx.w(24 + d);
#pragma unroll
    for (int r = 0; r < 12; r++) {
      const u64 o = nxt;
      if (r + 1 < 12) nxt = cx.w(24 + 2 * (r + 1) + d);
      // 12 terms with coefficients < 2^6 on the 32-bit halves: carry-free, one reduction (cf. SmallLin)
      SmallLin acc;
#pragma unroll
      for (int i = 0; i < 12; i++) acc.add(in[(i + r) % 12], (u32)poseidon::MDS_CIRC[i] + (i == 0 && r == 0 ? (u32)poseidon::MDS_DIAG0 : 0u));
      cx.at(2 * r + d, gl::sub(o, gl::canon(acc.value())));   // sub wants a canonical subtrahend
    }
  }
}
// upstream gates/coset_interpolation.rs eval_unfiltered_base_one (subgroup of order 16, degree 6, 2 intermediates):
// the point divided by the shift is given on wires and checked; the barycentric recurrence
//   eval' = eval (x - x_i) + w_i v_i prod,  prod' = prod (x - x_i)
// runs over the points in three chunks [0,6) [6,11) [11,16), its state pinned to wires between chunks.
__device__ void gate_coset_interp(Ctx& cx) {
  const u64 shift = cx.w(0);
  const gl::E2 point{cx.w(CI_W_POINT), cx.w(CI_W_POINT + 1)}, x{cx.w(CI_W_SHIFTED), cx.w(CI_W_SHIFTED + 1)};
  int nc = 0;
  cx.at(nc++, gl::sub(point.a, gl::mul(x.a, shift)));
  cx.at(nc++, gl::sub(point.b, gl::mul(x.b, shift)));
  const u64 g = gl::root_of_unity(4), inv16 = gl::inv(16);
  gl::E2 eval{0, 0}, prod{1, 0};
  u64 xi = 1;
  for (int c = 0; c <= CI_INTER; c++) {
    if (c > 0) {
      const gl::E2 ie{cx.w(CI_W_INTER + 2 * (c - 1)), cx.w(CI_W_INTER + 2 * (c - 1) + 1)};
      const gl::E2 ip{cx.w(CI_W_INTER + 2 * (CI_INTER + c - 1)), cx.w(CI_W_INTER + 2 * (CI_INTER + c - 1) + 1)};
      cx.at(nc++, gl::sub(ie.a, eval.a));
      cx.at(nc++, gl::sub(ie.b, eval.b));
      cx.at(nc++, gl::sub(ip.a, prod.a));
      cx.at(nc++, gl::sub(ip.b, prod.b));
      eval = ie;
      prod = ip;
    }
    const int end = c == 0 ? CI_DEGREE : (1 + (CI_DEGREE - 1) * (c + 1) < CI_POINTS ? 1 + (CI_DEGREE - 1) * (c + 1) : CI_POINTS);
    for (int i = c == 0 ? 0 : 1 + (CI_DEGREE - 1) * c; i < end; i++) {
      const gl::E2 v = gl::mul(gl::E2{cx.w(1 + 2 * i), cx.w(2 + 2 * i)}, gl::mul(xi, inv16));   // value * weight (= x_i / 16)
      const gl::E2 term{gl::sub(x.a, xi), x.b};
      eval = gl::add(gl::mul(eval, term), gl::mul(v, prod));
      prod = gl::mul(prod, term);
      xi = gl::mul(xi, g);
    }
  }
  cx.at(nc++, gl::sub(cx.w(CI_W_VALUE), eval.a));
  cx.at(nc++, gl::sub(cx.w(CI_W_VALUE + 1), eval.b));
}
// upstream gates/poseidon.rs eval_unfiltered_base_one (rounds in the defining form; same constraint polynomials as
// upstream's fast partial rounds, which are a linear change of basis on lanes 1..11)
__device__ void gate_poseidon(Ctx& cx) {
  int nc = 0;
  u64 swap = cx.w(24);
  cx.at(nc++, gl::mul_nc(swap, gl::sub(swap, 1)));
  u64 st[12];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u64 lhs = cx.w(i), rhs = cx.w(i + 4), delta = cx.w(25 + i);
    cx.at(nc++, gl::sub(gl::mul(swap, gl::sub(rhs, lhs)), delta));
    st[i] = gl::add(lhs, delta);
    st[i + 4] = gl::sub(rhs, delta);
  }
#pragma unroll
  for (int i = 8; i < 12; i++) st[i] = cx.w(i);
  int tr = 29;
  u64 nsb = cx.w(tr);   // the S-box input wire read one position ahead (wires 29 .. 29 + 117, in round order)
  constexpr int TR_END = 29 + 12 * (2 * poseidon::HALF_FULL - 1) + poseidon::N_PARTIAL;
  for (int r = 0; r < poseidon::N_ROUNDS; r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = poseidon::add_rc(st[i], poseidon::RC[12 * r + i]);
    if (r < poseidon::HALF_FULL || r >= poseidon::HALF_FULL + poseidon::N_PARTIAL) {
      if (r != 0) {
#pragma unroll
        for (int i = 0; i < 12; i++) {
          const u64 sb = nsb;
          tr++;
          if (tr < TR_END) nsb = cx.w(tr);
          cx.at(nc++, gl::sub(gl::canon(st[i]), sb));
          st[i] = sb;
        }
      }
#pragma unroll
      for (int i = 0; i < 12; i++) st[i] = poseidon::sbox(st[i]);
    } else {
      const u64 sb = nsb;
      tr++;
      if (tr < TR_END) nsb = cx.w(tr);
      cx.at(nc++, gl::sub(gl::canon(st[0]), sb));
      st[0] = poseidon::sbox(sb);
    }
    poseidon::mds(st);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) cx.at(nc++, gl::sub(gl::canon(st[i]), cx.w(12 + i)));
}

}  // namespace

// alpha_pows[c][j] = alpha_c^j, j < ALPHA_POWS
__global__ void k_alpha_pows(const u64* __restrict__ chal, u64* __restrict__ out) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  int c = blockIdx.x, j = threadIdx.x;
  if (j < ALPHA_POWS) out[c * ALPHA_POWS + j] = gl::pow(chal[CH_ALPHAS + c], (u64)j);
}
void launch_alpha_pows(const u64* d_chal, u64* d_alpha_pows, hipStream_t st) {
  hipLaunchKernelGGL(k_alpha_pows, dim3(2), dim3(ALPHA_POWS), 0, st, d_chal, d_alpha_pows);
}

// Occupancy target: unconstrained the kernel takes 255 VGPRs = 2 waves per SIMD, which fills the register
// file, leaves the VALU half idle on memory latency and lets no other stream's waves co-reside.  Capped at
// 96 VGPRs (5 waves per SIMD) the compiler spills ~580 B per lane to scratch, yet the kernel runs 2.49 ->
// 1.78 ms and the batch 110 -> 119 proofs/s (3, 4, 6 waves: 2.15, 1.85, 1.77 ms).  Round 3 (wires read one iteration
// ahead, wave priority 2): 4 / 5 / 6 waves measure 137.3 / 136.7 / 135.8 proofs/s -- with its loads overlapped the kernel
// no longer needs the fifth wave, and at 128 VGPRs it spills less.
constexpr int Q_WAVES = 4;
// The recursion instantiation (k_quotient_rec: aggregation circuits) is held to 3 waves per SIMD; the merged pass over the
// routed wires (below) is compiled in for it too.
constexpr int QREC_WAVES = 3;
// prefetch distance of the permutation-argument pass (1, 2, 4 and 8 were measured: profiles/r04_ab_quotient_variants.txt)
constexpr int Q_PF = 1;
// REC: the gate set of recursive-verifier circuits (adds ArithmeticExtensionGate and PoseidonGate).  The fib-64 hot
// path runs the REC = false instantiation, whose code is what it was before those gates existed.
template <bool REC>
__device__ __forceinline__ void quotient_body(const QuotientArgs& a) {
  __shared__ u64 ap[2 * ALPHA_POWS];
  __shared__ AlphaLimbs apl[ALPHA_POWS];
  for (int i = threadIdx.x; i < 2 * ALPHA_POWS; i += blockDim.x) {
    const u64 v = a.alpha_pows[i];
    ap[i] = v;
    const int ch = i / ALPHA_POWS, j = i - ch * ALPHA_POWS;
    apl[j].l[4 * ch] = (u32)v & 0x3FFFFFu;
    apl[j].l[4 * ch + 1] = (u32)(v >> 22) & 0x3FFFFFu;
    apl[j].l[4 * ch + 2] = (u32)(v >> 44);
    apl[j].l[4 * ch + 3] = 0;
  }
  __shared__ u64 kb[2 * MAX_ROUTED];  // k_j * beta_c
  for (int i = threadIdx.x; i < 2 * (int)a.num_routed; i += blockDim.x) {
    const int c = i / (int)a.num_routed, j = i - c * (int)a.num_routed;
    kb[i] = gl::mul(a.k_is[j], a.chal[CH_BETAS + c]);
  }
  __syncthreads();
  const uint32_t lde_bits = a.degree_bits + a.rate_bits;
  const size_t big = (size_t)1 << lde_bits;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // bit-reversed position
  if (p >= big) return;
  const uint32_t i_nat = gl::bitrev((u32)p, lde_bits);
  const u64 x = gl::mul(gl::GENERATOR, a.pow_big[i_nat]);
  const uint32_t rate_mask = (1u << a.rate_bits) - 1;
  const u64 zhx = a.zh[i_nat & rate_mask], zhi = a.zh_inv[i_nat & rate_mask];
  // position of the "next row" point: natural index + 2^rate_bits
  const uint32_t i_next = (i_nat + (1u << a.rate_bits)) & (uint32_t)(big - 1);
  const size_t p_next = gl::bitrev(i_next, lde_bits);

  const int NC = 2, NP = (int)a.num_partial_products, RW = (int)a.num_routed;
  // chunks of max_quotient_degree_factor routed wires (upstream partial_products.rs), NP + 1 of them
  const int nch = NP + 1, per = (int)a.quotient_degree_factor;
  const u32 boff = (u32)p * 8u;
  const LaneCols csc{a.cs_lde, boff, big}, wrc{a.wires_lde, boff, big}, zsc{a.zs_lde, boff, big};
  const int n_consts = (int)a.num_selectors;  // selectors first, then the 2 gate constants

  u64 res[2] = {0, 0};
  // --- L_0(x) (Z_c(x) - 1): terms 0..NC
  {
    u64 l0 = gl::mul_nc(zhx, a.l0_inv[p]);  // 1 / (n (x - 1)), per circuit (k_l0_inv)
    for (int c = 0; c < NC; c++) {
      u64 t = gl::mul_nc(l0, gl::sub_c(zsc.col(c), 1));
      res[0] = gl::mad_nc(t, ap[c], res[0]);          // res: any u64 from here on, canonical again in the final product
      res[1] = gl::mad_nc(t, ap[ALPHA_POWS + c], res[1]);
    }
  }
  auto gate_filter = [&](uint32_t gi) {
    const GateEntry ge = a.gates[gi];
    const u64 s = csc.col(ge.selector_index);
    u64 filter = 1;
    for (uint32_t k = ge.group_start; k < ge.group_end; k++)
      if (k != gi) filter = gl::mul_nc(filter, gl::sub_c((u64)k, s));
    if (a.num_selectors > 1) filter = gl::mul_nc(filter, gl::sub_c(0xFFFFFFFFull, s));
    return filter;
  };
  // --- partial-product checks: terms NC + c*nch + k.  Both challenges in ONE pass over the routed wires and their
  // sigmas: every column is read once instead of once per challenge (160 of the kernel's ~850 column reads).
  // FAST (the standard configuration: 80 routed wires): ONE more pass over the routed wires evaluates all the gates whose
  // wires lie in that range and are read in ascending order -- Constant, PublicInput, BaseSum, Arithmetic, MulExtension:
  // 80 column reads instead of their 228.  (Doing it inside the permutation pass itself was measured too: one read less
  // per column, but 192 spilled registers instead of 116 and no gain.)  Those gates share ONE alpha-fold accumulator: each
  // constraint is multiplied by its gate's filter first (116 multiplications, paid for by the five per-gate folds of the
  // accumulator limbs that are no longer needed);  sum_g f_g sum_j alpha^j c_gj = sum_j alpha^j sum_g f_g c_gj.
  const bool fast = per == 8 && RW == 80 && NP == 9 && a.num_wires >= 80;
  uint32_t merged_mask = 0;   // gate kinds evaluated by the merged pass
  u64 mg0 = 0, mg1 = 0;       // their filtered, alpha-folded sums
  const u64 beta0 = a.chal[CH_BETAS], beta1 = a.chal[CH_BETAS + 1];
  const u64 gamma0 = a.chal[CH_GAMMAS], gamma1 = a.chal[CH_GAMMAS + 1];
  if (fast) {
    for (uint32_t gi = 0; gi < a.n_gates; gi++) {
      const uint32_t kind = a.gates[gi].kind;
      if (kind == G_CONSTANT || kind == G_PUBLIC_INPUT || kind == G_BASE_SUM || kind == G_ARITHMETIC || kind == G_MUL_EXT) {
        if ((merged_mask >> kind) & 1u) { merged_mask = 0xFFFFFFFFu; break; }   // a kind twice: not this path
        merged_mask |= 1u << kind;
      }
    }
  }
  if (fast && merged_mask != 0xFFFFFFFFu) {
    const bool h_const = (merged_mask >> G_CONSTANT) & 1u, h_pi = (merged_mask >> G_PUBLIC_INPUT) & 1u,
               h_bsum = (merged_mask >> G_BASE_SUM) & 1u, h_arith = (merged_mask >> G_ARITHMETIC) & 1u,
               h_mext = (merged_mask >> G_MUL_EXT) & 1u;
    u64 f_const = 0, f_pi = 0, f_bsum = 0, f_arith = 0, f_mext = 0;
    for (uint32_t gi = 0; gi < a.n_gates; gi++) {
      const uint32_t kind = a.gates[gi].kind;
      if (kind == G_CONSTANT) f_const = gate_filter(gi);
      else if (kind == G_PUBLIC_INPUT) f_pi = gate_filter(gi);
      else if (kind == G_BASE_SUM) f_bsum = gate_filter(gi);
      else if (kind == G_ARITHMETIC) f_arith = gate_filter(gi);
      else if (kind == G_MUL_EXT) f_mext = gate_filter(gi);
    }
    const u64 k0 = csc.col(n_consts), k1 = csc.col(n_consts + 1);
    Ctx mc;          // the shared accumulator of the merged gates
    mc.wires = wrc;
    mc.big = big;
    mc.apl = apl;
    mc.reset();
    u64 bs_sum = 0, bs_w0 = 0;
    u64 wnext = wrc.col(0);   // the wire read one position ahead
    SmallLin bs_acc;
    u64 ar0 = 0, ar1 = 0, ar2 = 0;                    // arithmetic: multiplicand 0, multiplicand 1, addend of the current op
    u64 mx0 = 0, mx1 = 0, mx2 = 0, mx3 = 0, mx4 = 0;  // mul-extension: a, b and output.a of the current op
    // one routed wire: position JJ (compile time) of a block that starts at wire `base` (a multiple of 24, or 72)
#define P25_Q_WIRE(JJ)                                                                                          \
  {                                                                                                             \
    const int j = base + (JJ);                                                                                  \
    const u64 w = wnext;                                                                                        \
    if (j + 1 < 80) wnext = wrc.col(j + 1);                                                                      \
    if ((JJ) < 4 && base == 0) {                                                                                \
      if (h_const && (JJ) < 2) mc.at((JJ), gl::mul_nc(gl::sub_c((JJ) == 0 ? k0 : k1, w), f_const));            \
      if (h_pi) mc.at((JJ), gl::mul_nc(gl::sub_c(w, a.pi_hash[(JJ)]), f_pi));                                   \
    }                                                                                                           \
    if (h_bsum) {                                                                                               \
      if (j == 0) bs_w0 = w;                                                                                    \
      if (j >= 1 && j <= BASE_SUM_LIMBS) {                                                                      \
        constexpr int t = ((JJ) + 7) % 8;   /* limb i = j - 1, position i mod 8 of its group of eight */          \
        bs_acc.add(w, 1u << t);                                                                                 \
        mc.at(j, gl::mul_nc(bool01(w), f_bsum));                                                                \
        if (t == 7 || j == BASE_SUM_LIMBS) {                                                                    \
          bs_sum = gl::mad_nc(bs_acc.value(), (u64)1 << (8 * (((j - 1) >> 3) & 7)), bs_sum);                          \
          bs_acc = SmallLin();                                                                                  \
        }                                                                                                       \
      }                                                                                                         \
    }                                                                                                           \
    if (h_arith) {                                                                                              \
      if ((JJ) % 4 == 0) ar0 = w;                                                                               \
      else if ((JJ) % 4 == 1) ar1 = w;                                                                          \
      else if ((JJ) % 4 == 2) ar2 = w;                                                                          \
      else {                                                                                                    \
        const u64 comp = gl::mad_nc(gl::mul_nc(ar0, ar1), k0, gl::mul_nc(ar2, k1));                             \
        mc.at(j >> 2, gl::mul_nc(gl::sub_nc(w, comp), f_arith));                                                \
      }                                                                                                         \
    }                                                                                                           \
    if (h_mext && j < 78) {                                                                                     \
      if ((JJ) % 6 == 0) mx0 = w;                                                                               \
      else if ((JJ) % 6 == 1) mx1 = w;                                                                          \
      else if ((JJ) % 6 == 2) mx2 = w;                                                                          \
      else if ((JJ) % 6 == 3) mx3 = w;                                                                          \
      else if ((JJ) % 6 == 4) mx4 = w;                                                                          \
      else {                                                                                                    \
        const gl::E2 pr = e2_mul_scaled(gl::E2{mx0, mx1}, gl::E2{mx2, mx3}, k0);                                \
        const int op = j / 6;                                                                                   \
        mc.at(2 * op, gl::mul_nc(gl::sub_nc(mx4, pr.a), f_mext));                                               \
        mc.at(2 * op + 1, gl::mul_nc(gl::sub_nc(w, pr.b), f_mext));                                             \
      }                                                                                                         \
    }                                                                                                           \
  }
#define P25_Q_WIRES8(J0) P25_Q_WIRE(J0) P25_Q_WIRE(J0 + 1) P25_Q_WIRE(J0 + 2) P25_Q_WIRE(J0 + 3) \
                         P25_Q_WIRE(J0 + 4) P25_Q_WIRE(J0 + 5) P25_Q_WIRE(J0 + 6) P25_Q_WIRE(J0 + 7)
    for (int base = 0; base < 72; base += 24) {   // 24 = lcm of the op sizes 4 (arithmetic), 6 (mul-ext), 8 (chunks, limb groups)
      P25_Q_WIRES8(0) P25_Q_WIRES8(8) P25_Q_WIRES8(16)
    }
    {
      const int base = 72;
      P25_Q_WIRES8(0)
    }
#undef P25_Q_WIRES8
#undef P25_Q_WIRE
    if (h_bsum) mc.at(0, gl::mul_nc(gl::sub_c(bs_sum, bs_w0), f_bsum));
    mg0 = mc.acc0();
    mg1 = mc.acc1();
  } else {
    merged_mask = 0;
  }
  {
    // wires and sigmas read Q_PF positions ahead (a ring of that many register pairs; the chunk length, 8, is a
    // multiple of it, so the ring index is static in the unrolled loop)
    constexpr int PF = Q_PF;
    u64 wq[PF], sq[PF];
#pragma unroll
    for (int d = 0; d < PF; d++) {
      wq[d] = d < RW ? wrc.col(d) : 0;
      sq[d] = d < RW ? csc.col(n_consts + 2 + d) : 0;
    }
    for (int k = 0; k < nch; k++) {
      u64 np0 = 1, dp0 = 1, np1 = 1, dp1 = 1;
      // beta * k_j * x: k_j * beta comes from a per-proof table (k_alpha_pows fills it)
      for (int j0 = k * per; j0 < (k + 1) * per && j0 < RW; j0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; d++) {
          const int j = j0 + d;
          if (j >= RW || j >= (k + 1) * per) break;
          const u64 w = wq[d], sg = sq[d];
          if (j + PF < RW) {
            wq[d] = wrc.col(j + PF);
            sq[d] = csc.col(n_consts + 2 + j + PF);
          }
          const u64 wg0 = gl::add_c(w, gamma0), wg1 = gl::add_c(w, gamma1);
          np0 = gl::mul_nc(np0, gl::mad_nc(kb[j], x, wg0));
          dp0 = gl::mul_nc(dp0, gl::mad_nc(beta0, sg, wg0));
          np1 = gl::mul_nc(np1, gl::mad_nc(kb[RW + j], x, wg1));
          dp1 = gl::mul_nc(dp1, gl::mad_nc(beta1, sg, wg1));
        }
      }
#pragma unroll
      for (int c = 0; c < NC; c++) {
        const u64 np = c ? np1 : np0, dp = c ? dp1 : dp0;
        u64 prev = k == 0 ? zsc.col(c) : zsc.col(NC + c * NP + k - 1);
        u64 next = k == NP ? a.zs_lde[(size_t)c * big + p_next] : zsc.col(NC + c * NP + k);
        u64 t = gl::sub_nc(gl::mul_nc(prev, np), gl::mul_nc(next, dp));
        int ti = NC + c * nch + k;
        res[0] = gl::mad_nc(t, ap[ti], res[0]);
        res[1] = gl::mad_nc(t, ap[ALPHA_POWS + ti], res[1]);
      }
    }
  }
  // --- gate constraints, terms NC*(1+nch) + j
  {
    Ctx cx;
    cx.wires = wrc;
    cx.big = big;
    cx.apl = apl;
    const u64 k0 = csc.col(n_consts), k1 = csc.col(n_consts + 1);
    u64 g0 = mg0, g1 = mg1;
    for (uint32_t gi = 0; gi < a.n_gates; gi++) {
      const GateEntry ge = a.gates[gi];
      if ((merged_mask >> ge.kind) & 1u) continue;   // evaluated by the merged pass above
      const u64 filter = gate_filter(gi);
      cx.reset();
      {
        // Column descriptors are loop-invariant and the compiler would hoist ALL of them (every column of every
        // evaluator) out of this loop, into hundreds of spilled scalar registers; an opaque copy of the stride per
        // iteration keeps each one next to its load.
        size_t bg = big;
        asm volatile("" : "+s"(bg));
        cx.wires.big = bg;
      }
      switch (ge.kind) {
        case G_CONSTANT: gate_constant(cx, k0, k1); break;
        case G_PUBLIC_INPUT: gate_public_input(cx, a.pi_hash); break;
        case G_BASE_SUM: gate_base_sum(cx); break;
        case G_U32_INTERLEAVE: gate_u32_interleave(cx); break;
        case G_U32_UNINTERLEAVE: gate_u32_uninterleave(cx); break;
        case G_ARITHMETIC: gate_arithmetic(cx, k0, k1); break;
        case G_MUL_EXT: gate_mul_ext(cx, k0); break;
        case G_EXPONENTIATION: gate_exponentiation(cx); break;
        case G_U32_ARITHMETIC: gate_u32_arithmetic(cx); break;
        case G_POSEIDON2: gate_poseidon2(cx); break;
        case G_ARITH_EXT:
          if constexpr (REC) gate_arith_ext(cx, k0, k1);
          break;
        case G_POSEIDON:
          if constexpr (REC) gate_poseidon(cx);
          break;
        case G_RANDOM_ACCESS:
          if constexpr (REC) gate_random_access(cx, k0, k1);
          break;
        case G_REDUCING:
          if constexpr (REC) gate_reducing<false>(cx);
          break;
        case G_REDUCING_EXT:
          if constexpr (REC) gate_reducing<true>(cx);
          break;
        case G_COSET_INTERP:
          if constexpr (REC) gate_coset_interp(cx);
          break;
        case G_POSEIDON_MDS:
          if constexpr (REC) gate_poseidon_mds(cx);
          break;
        default: break;  // NoopGate: no constraints
      }
      g0 = gl::mad_nc(filter, cx.acc0(), g0);
      g1 = gl::mad_nc(filter, cx.acc1(), g1);
    }
    const int off = NC * (1 + nch);
    res[0] = gl::mad_nc(g0, ap[off], res[0]);
    res[1] = gl::mad_nc(g1, ap[ALPHA_POWS + off], res[1]);
  }
  a.out[p] = gl::mul(res[0], zhi);
  a.out[big + p] = gl::mul(res[1], zhi);
}

__global__ __launch_bounds__(128, Q_WAVES) void k_quotient(QuotientArgs a) {
  P25_WAVE_PRIO(P25_PRIO_BULK); quotient_body<false>(a); }
__global__ __launch_bounds__(128, QREC_WAVES) void k_quotient_rec(QuotientArgs a) {
  P25_WAVE_PRIO(P25_PRIO_BULK); quotient_body<true>(a); }

// out[p] = 1 / (n (x_p - 1)), x_p = g w_big^rev(p): the point-dependent factor of L_0(x) = Z_H(x) / (n (x - 1)).
__global__ __launch_bounds__(256) void k_l0_inv(const u64* __restrict__ pow_big, uint32_t degree_bits,
                                                uint32_t lde_bits, u64* __restrict__ out) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >> lde_bits) return;
  const u64 x = gl::mul(gl::GENERATOR, pow_big[gl::bitrev((u32)p, lde_bits)]);
  out[p] = gl::inv(gl::mul((u64)1 << degree_bits, gl::sub(x, 1)));
}
void launch_l0_inv(const u64* d_pow_big, uint32_t degree_bits, uint32_t rate_bits, u64* d_out, hipStream_t st) {
  const size_t big = (size_t)1 << (degree_bits + rate_bits);
  hipLaunchKernelGGL(k_l0_inv, dim3((unsigned)((big + 255) / 256)), dim3(256), 0, st, d_pow_big, degree_bits,
                     degree_bits + rate_bits, d_out);
}

void launch_quotient(const QuotientArgs& a, hipStream_t st) {
  if (a.num_routed > (uint32_t)MAX_ROUTED) throw std::runtime_error("quotient: more than MAX_ROUTED routed wires");
  for (uint32_t gi = 0; gi < a.n_gates; gi++)  // the carry-free alpha fold holds 512 terms per gate
    if (gate_info((GateKind)a.gates[gi].kind).num_constraints > MAX_TERMS_PER_FOLD ||
        gate_info((GateKind)a.gates[gi].kind).num_constraints > ALPHA_POWS)
      throw std::runtime_error("quotient: gate with too many constraints for the alpha-power table");
  const size_t big = (size_t)1 << (a.degree_bits + a.rate_bits);
  bool rec = false;
  for (uint32_t gi = 0; gi < a.n_gates; gi++) rec |= a.gates[gi].kind >= G_ARITH_EXT;   // the recursion gate set
  const unsigned grid = (unsigned)((big + 127) / 128);
  if (rec)
    hipLaunchKernelGGL(k_quotient_rec, dim3(grid), dim3(128), 0, st, a);
  else
    hipLaunchKernelGGL(k_quotient, dim3(grid), dim3(128), 0, st, a);
}

}  // namespace p25
