// Poseidon partial rounds, three at a time (device only; included by poseidon.h).
//
// In a partial round only lane 0 passes through the S-box, so with v the state entering the round
// (constants added), d = sbox(v0) - v0 and m0 = column 0 of the MDS matrix M:
//     v' = M v + d m0 + c'.
// Three rounds compose to
//     v3 = M^3 v + d0 (M^2 m0) + d1 (M m0) + d2 m0 + (M^2 c1 + M c2 + c3),
// and the two intermediate S-box inputs need one matrix ROW each:
//     v1[0] = (M v)[0] + d0 m0[0] + c1[0],     v2[0] = (M^2 v)[0] + d0 (M m0)[0] + d1 m0[0] + (M c1 + c2)[0].
// M has entries <= 41, so M^3 has entries < 2^22 and row sums 264^3 < 2^24.2: on 32-bit halves of the
// state every sum still fits a 64-bit accumulator (< 2^57), i.e. the same carry-free v_mad_u64_u32
// accumulation as the one-round layer (mds_rc) -- (12 + 3) terms x 2 halves per output word once per
// THREE rounds, plus two single rows: 442 multiply-adds instead of 3 x 288.  (Four rounds would
// overflow the accumulators; upstream's sparse "fast partial rounds" need 64-bit constants, i.e. full
// modular multiplies, and come out no cheaper than the dense layer on this ISA.)
// The arithmetic is exact integer arithmetic mod p: results are bit-identical to the round-by-round form.
#pragma once

namespace p3r {

constexpr int BLOCKS = 7;  // 21 of the 22 partial rounds; the last one runs through mds_rc

struct Row16 {
  u32 c[16];
};
struct Tables {
  Row16 t3[WIDTH];        // M^3 row r | (M^2 m0)[r], (M m0)[r], m0[r], 0
  Row16 r2;               // row 0 of M^2 | (M m0)[0], m0[0], 0, 0
  u32 kc[BLOCKS][64];     // per block: k1.lo, k1.hi, k2.lo, k2.hi, then K3[r].lo, K3[r].hi (r < 12), padding
};

constexpr u64 mulmod(u64 a, u64 b) { return (u64)((unsigned __int128)a * b % gl::P); }
constexpr u64 addmod(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % gl::P); }

constexpr Tables make_tables() {
  Tables t{};
  u64 M[WIDTH][WIDTH] = {}, M2[WIDTH][WIDTH] = {}, M3[WIDTH][WIDTH] = {};
  for (int r = 0; r < WIDTH; r++)
    for (int j = 0; j < WIDTH; j++) M[r][j] = MDS_CIRC[(j - r + WIDTH) % WIDTH] + (r == 0 && j == 0 ? MDS_DIAG0 : 0);
  for (int r = 0; r < WIDTH; r++)
    for (int j = 0; j < WIDTH; j++)
      for (int k = 0; k < WIDTH; k++) M2[r][j] += M[r][k] * M[k][j];
  for (int r = 0; r < WIDTH; r++)
    for (int j = 0; j < WIDTH; j++)
      for (int k = 0; k < WIDTH; k++) M3[r][j] += M2[r][k] * M[k][j];
  // v' = M (v + d e0) + c' = M v + d (M e0): m0 = column 0 of M, M m0 = column 0 of M^2, M^2 m0 = column 0 of M^3
  for (int r = 0; r < WIDTH; r++) {
    for (int j = 0; j < WIDTH; j++) t.t3[r].c[j] = (u32)M3[r][j];
    t.t3[r].c[12] = (u32)M3[r][0];
    t.t3[r].c[13] = (u32)M2[r][0];
    t.t3[r].c[14] = (u32)M[r][0];
    t.t3[r].c[15] = 0;
  }
  for (int j = 0; j < WIDTH; j++) t.r2.c[j] = (u32)M2[0][j];
  t.r2.c[12] = (u32)M2[0][0];
  t.r2.c[13] = (u32)M[0][0];
  for (int b = 0; b < BLOCKS; b++) {
    const int r0 = HALF_FULL + 3 * b;
    const u64* c1 = RC + WIDTH * (r0 + 1);
    const u64* c2 = RC + WIDTH * (r0 + 2);
    const u64* c3 = RC + WIDTH * (r0 + 3);
    u64 k1 = c1[0];
    u64 k2 = c2[0];
    for (int j = 0; j < WIDTH; j++) k2 = addmod(k2, mulmod(M[0][j], c1[j]));
    t.kc[b][0] = (u32)k1;
    t.kc[b][1] = (u32)(k1 >> 32);
    t.kc[b][2] = (u32)k2;
    t.kc[b][3] = (u32)(k2 >> 32);
    for (int r = 0; r < WIDTH; r++) {
      u64 k3 = c3[r];
      for (int j = 0; j < WIDTH; j++) k3 = addmod(k3, addmod(mulmod(M2[r][j], c1[j]), mulmod(M[r][j], c2[j])));
      t.kc[b][4 + 2 * r] = (u32)k3;
      t.kc[b][5 + 2 * r] = (u32)(k3 >> 32);
    }
  }
  return t;
}
static constexpr Tables TBL = make_tables();

// Compile-time check of the algebra: three rounds through the tables == three rounds one at a time
// (plain modular arithmetic, on a fixed non-trivial state, for every block).
constexpr u64 sbox_ref(u64 x) {
  u64 x2 = mulmod(x, x), x4 = mulmod(x2, x2), x3 = mulmod(x, x2);
  return mulmod(x3, x4);
}
constexpr bool tables_consistent() {
  for (int b = 0; b < BLOCKS; b++) {
    u64 v[WIDTH] = {}, w[WIDTH] = {};
    for (int i = 0; i < WIDTH; i++) v[i] = w[i] = mulmod(0x9E3779B97F4A7C15ull % gl::P, (u64)(i + 1 + 13 * b)) ^ 0;
    // reference: round by round, constants of the following round added after the MDS layer
    const int r0 = HALF_FULL + 3 * b;
    for (int k = 0; k < 3; k++) {
      u64 y[WIDTH] = {};
      for (int i = 0; i < WIDTH; i++) y[i] = w[i];
      y[0] = sbox_ref(y[0]);
      for (int r = 0; r < WIDTH; r++) {
        u64 acc = RC[WIDTH * (r0 + k + 1) + r];
        for (int j = 0; j < WIDTH; j++)
          acc = addmod(acc, mulmod(MDS_CIRC[(j - r + WIDTH) % WIDTH] + (r == 0 && j == 0 ? MDS_DIAG0 : 0), y[j]));
        w[r] = acc;
      }
    }
    // through the tables
    u64 d[3] = {};
    u64 x = v[0];
    d[0] = addmod(sbox_ref(x), gl::P - x);
    const u64 row0[WIDTH] = {25, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    u64 a = (u64)TBL.kc[b][0] | ((u64)TBL.kc[b][1] << 32);
    for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(row0[j], v[j]));
    a = addmod(a, mulmod(25, d[0]));
    x = a;
    d[1] = addmod(sbox_ref(x), gl::P - x);
    a = (u64)TBL.kc[b][2] | ((u64)TBL.kc[b][3] << 32);
    for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(TBL.r2.c[j], v[j]));
    a = addmod(a, addmod(mulmod(TBL.r2.c[12], d[0]), mulmod(TBL.r2.c[13], d[1])));
    x = a;
    d[2] = addmod(sbox_ref(x), gl::P - x);
    for (int r = 0; r < WIDTH; r++) {
      a = (u64)TBL.kc[b][4 + 2 * r] | ((u64)TBL.kc[b][5 + 2 * r] << 32);
      for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(TBL.t3[r].c[j], v[j]));
      for (int e = 0; e < 3; e++) a = addmod(a, mulmod(TBL.t3[r].c[12 + e], d[e]));
      if (a != w[r]) return false;
    }
  }
  return true;
}
static_assert(tables_consistent(), "poseidon_p3r.h: three-round tables disagree with the round-by-round definition");

typedef const Tables __attribute__((address_space(4))) * tbl_ptr;

__device__ __forceinline__ void mad_s(u64& acc, u32 x, u32 coef_sgpr) {
  u64 dm;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(dm) : "v"(x), "s"(coef_sgpr));
}
__device__ __forceinline__ u64 mul_s(u32 x, u32 coef_sgpr) {
  u64 acc, dm;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(acc), "=s"(dm) : "v"(x), "s"(coef_sgpr));
  return acc;
}
__device__ __forceinline__ void add_s(u64& acc, u32 k_sgpr) {  // acc += k (a zero-extended 32-bit constant)
  u64 dm;
  asm("v_mad_u64_u32 %0, %1, 1, %2, %0" : "+v"(acc), "=s"(dm) : "s"(k_sgpr));
}
// al + ah * 2^32 mod p for al, ah < 2^58 (non-canonical result); same sequence as in mds_rc
__device__ __forceinline__ u64 reduce_row(u64 al, u64 ah) {
  u32 ahl = (u32)ah, ahh = (u32)(ah >> 32);
  u64 X, dm, t;
  asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(X), "=s"(dm) : "v"(ahh), "v"(al));
  u32 x0 = (u32)X, x1 = (u32)(X >> 32);
  asm("v_add_co_u32_e32 %1, vcc, %1, %3\n\ts_nop 1\n\t"
      "v_subbrev_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
      "s_andn2_b64 %2, vcc, %2\n\t"
      "v_addc_co_u32_e64 %1, %2, 0, %1, %2"
      : "+v"(x0), "+v"(x1), "=&s"(t)
      : "v"(ahl)
      : "vcc", "scc");
  return gl::make64(x0, x1);
}

// One row: sum_j coef[j] v_j + sum_e coef[12 + e] d_e + k, on 32-bit halves.
template <int NEXTRA>
__device__ __forceinline__ u64 row(const u32* lo, const u32* hi, const Row16& cf, const u32* dlo, const u32* dhi,
                                   u32 klo, u32 khi) {
  u64 al = mul_s(lo[0], cf.c[0]), ah = mul_s(hi[0], cf.c[0]);
#pragma unroll
  for (int j = 1; j < WIDTH; j++) {
    mad_s(al, lo[j], cf.c[j]);
    mad_s(ah, hi[j], cf.c[j]);
  }
#pragma unroll
  for (int e = 0; e < NEXTRA; e++) {
    mad_s(al, dlo[e], cf.c[12 + e]);
    mad_s(ah, dhi[e], cf.c[12 + e]);
  }
  add_s(al, klo);
  add_s(ah, khi);
  return reduce_row(al, ah);
}

// s: state entering partial round HALF_FULL + 3 b (constants added, any u64 representatives);
// on return: state entering round HALF_FULL + 3 b + 3.
__device__ __forceinline__ void three_rounds(u64 s[WIDTH], tbl_ptr tp, int b) {
  u32 lo[WIDTH], hi[WIDTH];
#pragma unroll
  for (int i = 0; i < WIDTH; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  const auto* kc = tp->kc[b];
  u32 dlo[3], dhi[3];
  u64 x = gl::canon(s[0]);
  u64 d = gl::sub(sbox(x), x);  // sbox(x) - x: any u64 minus a canonical value
  dlo[0] = (u32)d;
  dhi[0] = (u32)(d >> 32);
  {
    // row 0 of M has inline-constant entries, but one code path for all rows keeps this short: M row 0 =
    // (25, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20), m0[0] = 25
    u64 al = 0, ah = 0;
    mad_k<25>(al, lo[0]);  mad_k<25>(ah, hi[0]);
    mad_k<15>(al, lo[1]);  mad_k<15>(ah, hi[1]);
    mad_k<41>(al, lo[2]);  mad_k<41>(ah, hi[2]);
    mad_k<16>(al, lo[3]);  mad_k<16>(ah, hi[3]);
    mad_k<2>(al, lo[4]);   mad_k<2>(ah, hi[4]);
    mad_k<28>(al, lo[5]);  mad_k<28>(ah, hi[5]);
    mad_k<13>(al, lo[6]);  mad_k<13>(ah, hi[6]);
    mad_k<13>(al, lo[7]);  mad_k<13>(ah, hi[7]);
    mad_k<39>(al, lo[8]);  mad_k<39>(ah, hi[8]);
    mad_k<18>(al, lo[9]);  mad_k<18>(ah, hi[9]);
    mad_k<34>(al, lo[10]); mad_k<34>(ah, hi[10]);
    mad_k<20>(al, lo[11]); mad_k<20>(ah, hi[11]);
    mad_k<25>(al, dlo[0]); mad_k<25>(ah, dhi[0]);
    add_s(al, kc[0]);
    add_s(ah, kc[1]);
    x = gl::canon(reduce_row(al, ah));
  }
  d = gl::sub(sbox(x), x);
  dlo[1] = (u32)d;
  dhi[1] = (u32)(d >> 32);
  {
    Row16 cf;
#pragma unroll
    for (int j = 0; j < 16; j++) cf.c[j] = tp->r2.c[j];
    x = gl::canon(row<2>(lo, hi, cf, dlo, dhi, kc[2], kc[3]));
  }
  d = gl::sub(sbox(x), x);
  dlo[2] = (u32)d;
  dhi[2] = (u32)(d >> 32);
#pragma unroll
  for (int r = 0; r < WIDTH; r++) {
    Row16 cf;
#pragma unroll
    for (int j = 0; j < 16; j++) cf.c[j] = tp->t3[r].c[j];
    s[r] = row<3>(lo, hi, cf, dlo, dhi, kc[4 + 2 * r], kc[5 + 2 * r]);
  }
}

}  // namespace p3r
