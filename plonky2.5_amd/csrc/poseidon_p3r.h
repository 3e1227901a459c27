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
//
// The 22 partial rounds (4..25) are scheduled as
//   * a HEAD block that starts from the S-box outputs y of full round 3: its first "round" is just that
//     round's MDS layer (the same formulas with d0 = 0), followed by partial rounds 4 and 5;
//   * six regular blocks for rounds 6..23;
//   * a TAIL block of two rounds (24, 25) built on M^2.
// so the MDS layer of full round 3 costs nothing extra and no single round is left over.
// Round constants of words 1..11 are DEFERRED: between the head block and the tail block the registers hold
// t = (true state) - o, o a constant vector with o[0] = 0 that each block pushes through its own linear map
// (o' = M^3 o + K3, its word 0 taken out and added to the block's row 0), so that the head and the six regular blocks
// add constants to the two S-box inputs and to row 0 only -- 22 multiply-adds per block less -- and the tail block's
// twelve constants absorb M^2 o and make the state true again.  (Word 0 cannot be deferred: it goes through the S-box.)
// The arithmetic is exact integer arithmetic mod p: results are bit-identical to the round-by-round form.
#pragma once

namespace p3r {

constexpr int BLOCKS = 6;         // regular three-round blocks: rounds 6 + 3 b .. 8 + 3 b
constexpr int HEAD_ROUND = 3;     // the head block replaces MDS(round 3) + partial rounds 4, 5
constexpr int FIRST_BLOCK = 6;    // first round of regular block 0
constexpr int TAIL_ROUND = 24;    // the tail block covers partial rounds 24, 25
static_assert(HEAD_ROUND == HALF_FULL - 1 && FIRST_BLOCK + 3 * BLOCKS == TAIL_ROUND &&
                  TAIL_ROUND + 2 == HALF_FULL + N_PARTIAL,
              "partial-round schedule");

struct Row16 {
  u32 c[16];
};
struct Tables {
  Row16 t3[WIDTH];        // M^3 row r | (M^2 m0)[r], (M m0)[r], m0[r], 0
  Row16 t2[WIDTH];        // M^2 row r | (M m0)[r], m0[r], 0, 0
  Row16 r2;               // row 0 of M^2 | (M m0)[0], m0[0], 0, 0
  u32 kc[BLOCKS][64];     // per block: k1.lo, k1.hi, k2.lo, k2.hi, then K3[r].lo, K3[r].hi (r < 12), padding
  u32 kh[64];             // head block, same layout
  u32 kt[64];             // tail block: k1.lo, k1.hi, 0, 0, then K2[r].lo, K2[r].hi
};

constexpr u64 mulmod(u64 a, u64 b) { return (u64)((unsigned __int128)a * b % gl::P); }
constexpr u64 addmod(u64 a, u64 b) { return (u64)(((unsigned __int128)a + b) % gl::P); }
constexpr u64 mds_entry(int r, int j) { return MDS_CIRC[(j - r + WIDTH) % WIDTH] + (r == 0 && j == 0 ? MDS_DIAG0 : 0); }

// constants of a block whose three linear layers are followed by the constants of rounds r0+1, r0+2, r0+3, given the
// deferred offset o of the state entering it (o[0] == 0); on return o is the offset of the state leaving it
constexpr void block_constants(u32* out, int r0, const u64 (&M)[WIDTH][WIDTH], const u64 (&M2)[WIDTH][WIDTH],
                               const u64 (&M3)[WIDTH][WIDTH], u64 (&o)[WIDTH]) {
  const u64* c1 = RC + WIDTH * (r0 + 1);
  const u64* c2 = RC + WIDTH * (r0 + 2);
  const u64* c3 = RC + WIDTH * (r0 + 3);
  u64 k1 = c1[0];
  u64 k2 = c2[0];
  for (int j = 0; j < WIDTH; j++) {
    k1 = addmod(k1, mulmod(M[0][j], o[j]));
    k2 = addmod(k2, addmod(mulmod(M[0][j], c1[j]), mulmod(M2[0][j], o[j])));
  }
  out[0] = (u32)k1;
  out[1] = (u32)(k1 >> 32);
  out[2] = (u32)k2;
  out[3] = (u32)(k2 >> 32);
  u64 on[WIDTH] = {};
  for (int r = 0; r < WIDTH; r++) {
    u64 k3 = c3[r];
    for (int j = 0; j < WIDTH; j++)
      k3 = addmod(k3, addmod(addmod(mulmod(M2[r][j], c1[j]), mulmod(M[r][j], c2[j])), mulmod(M3[r][j], o[j])));
    on[r] = k3;
  }
  out[4] = (u32)on[0];          // row 0 takes its constant now (it is the next S-box input)
  out[5] = (u32)(on[0] >> 32);
  for (int r = 1; r < WIDTH; r++) {
    out[4 + 2 * r] = 0;          // rows 1..11: deferred
    out[5 + 2 * r] = 0;
  }
  on[0] = 0;
  for (int r = 0; r < WIDTH; r++) o[r] = on[r];
}

constexpr Tables make_tables() {
  Tables t{};
  u64 M[WIDTH][WIDTH] = {}, M2[WIDTH][WIDTH] = {}, M3[WIDTH][WIDTH] = {};
  for (int r = 0; r < WIDTH; r++)
    for (int j = 0; j < WIDTH; j++) M[r][j] = mds_entry(r, j);
  for (int r = 0; r < WIDTH; r++)
    for (int j = 0; j < WIDTH; j++)
      for (int k = 0; k < WIDTH; k++) M2[r][j] += M[r][k] * M[k][j];
  for (int r = 0; r < WIDTH; r++)
    for (int j = 0; j < WIDTH; j++)
      for (int k = 0; k < WIDTH; k++) M3[r][j] += M2[r][k] * M[k][j];
  // v' = M (v + d e0) + c' = M v + d (M e0): m0 = column 0 of M, M m0 = column 0 of M^2, M^2 m0 = column 0 of M^3
  for (int r = 0; r < WIDTH; r++) {
    for (int j = 0; j < WIDTH; j++) {
      t.t3[r].c[j] = (u32)M3[r][j];
      t.t2[r].c[j] = (u32)M2[r][j];
    }
    t.t3[r].c[12] = (u32)M3[r][0];
    t.t3[r].c[13] = (u32)M2[r][0];
    t.t3[r].c[14] = (u32)M[r][0];
    t.t2[r].c[12] = (u32)M2[r][0];
    t.t2[r].c[13] = (u32)M[r][0];
  }
  for (int j = 0; j < WIDTH; j++) t.r2.c[j] = (u32)M2[0][j];
  t.r2.c[12] = (u32)M2[0][0];
  t.r2.c[13] = (u32)M[0][0];
  u64 o[WIDTH] = {};   // the deferred offset, in program order: head block, regular blocks, tail
  block_constants(t.kh, HEAD_ROUND, M, M2, M3, o);
  for (int b = 0; b < BLOCKS; b++) block_constants(t.kc[b], FIRST_BLOCK + 3 * b, M, M2, M3, o);
  {  // tail: v2 = M^2 (t + o) + d0 (M m0) + d1 m0 + (M c1 + c2), c1 = RC[25], c2 = RC[26]: the state is true again
    const u64* c1 = RC + WIDTH * (TAIL_ROUND + 1);
    const u64* c2 = RC + WIDTH * (TAIL_ROUND + 2);
    u64 k1 = c1[0];
    for (int j = 0; j < WIDTH; j++) k1 = addmod(k1, mulmod(M[0][j], o[j]));
    t.kt[0] = (u32)k1;
    t.kt[1] = (u32)(k1 >> 32);
    for (int r = 0; r < WIDTH; r++) {
      u64 k = c2[r];
      for (int j = 0; j < WIDTH; j++) k = addmod(k, addmod(mulmod(M[r][j], c1[j]), mulmod(M2[r][j], o[j])));
      t.kt[4 + 2 * r] = (u32)k;
      t.kt[5 + 2 * r] = (u32)(k >> 32);
    }
  }
  return t;
}
static constexpr Tables TBL = make_tables();

// ---- compile-time check of the algebra: every block through the tables == the same rounds one at a time
// (plain modular arithmetic, on a fixed non-trivial state)
constexpr u64 sbox_ref(u64 x) {
  u64 x2 = mulmod(x, x), x4 = mulmod(x2, x2), x3 = mulmod(x, x2);
  return mulmod(x3, x4);
}
constexpr u64 kpair(const u32* k, int i) { return (u64)k[i] | ((u64)k[i + 1] << 32); }
// reference: `n` rounds starting at round r0 (state has RC[r0] added); skip_first: the first round has no S-box
constexpr void rounds_ref(u64 (&w)[WIDTH], int r0, int n, bool skip_first) {
  for (int k = 0; k < n; k++) {
    u64 y[WIDTH] = {};
    for (int i = 0; i < WIDTH; i++) y[i] = w[i];
    if (!(skip_first && k == 0)) y[0] = sbox_ref(y[0]);
    for (int r = 0; r < WIDTH; r++) {
      u64 acc = RC[WIDTH * (r0 + k + 1) + r];
      for (int j = 0; j < WIDTH; j++) acc = addmod(acc, mulmod(mds_entry(r, j), y[j]));
      w[r] = acc;
    }
  }
}
// one block through the tables, exactly as three_rounds does it (rows 1..11 without a constant)
constexpr void block_emulated(u64 (&v)[WIDTH], const u32* kc, bool skip_first) {
  u64 d[3] = {};
  u64 x = v[0];
  d[0] = skip_first ? 0 : addmod(sbox_ref(x), gl::P - x);
  u64 a = kpair(kc, 0);
  for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(mds_entry(0, j), v[j]));
  a = addmod(a, mulmod(mds_entry(0, 0), d[0]));
  x = a;
  d[1] = addmod(sbox_ref(x), gl::P - x);
  a = kpair(kc, 2);
  for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(TBL.r2.c[j], v[j]));
  a = addmod(a, addmod(mulmod(TBL.r2.c[12], d[0]), mulmod(TBL.r2.c[13], d[1])));
  x = a;
  d[2] = addmod(sbox_ref(x), gl::P - x);
  u64 w[WIDTH] = {};
  for (int r = 0; r < WIDTH; r++) {
    a = r == 0 ? kpair(kc, 4) : 0;
    for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(TBL.t3[r].c[j], v[j]));
    for (int e = 0; e < 3; e++) a = addmod(a, mulmod(TBL.t3[r].c[12 + e], d[e]));
    w[r] = a;
  }
  for (int r = 0; r < WIDTH; r++) v[r] = w[r];
}
constexpr void tail_emulated(u64 (&v)[WIDTH]) {
  u64 d[2] = {};
  u64 x = v[0];
  d[0] = addmod(sbox_ref(x), gl::P - x);
  u64 a = kpair(TBL.kt, 0);
  for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(mds_entry(0, j), v[j]));
  a = addmod(a, mulmod(mds_entry(0, 0), d[0]));
  x = a;
  d[1] = addmod(sbox_ref(x), gl::P - x);
  u64 w[WIDTH] = {};
  for (int r = 0; r < WIDTH; r++) {
    a = kpair(TBL.kt, 4 + 2 * r);
    for (int j = 0; j < WIDTH; j++) a = addmod(a, mulmod(TBL.t2[r].c[j], v[j]));
    for (int e = 0; e < 2; e++) a = addmod(a, mulmod(TBL.t2[r].c[12 + e], d[e]));
    w[r] = a;
  }
  for (int r = 0; r < WIDTH; r++) v[r] = w[r];
}
// the whole chain (MDS of round 3 + the 22 partial rounds) through the tables == the same rounds one at a time
constexpr bool chain_consistent(int salt) {
  u64 v[WIDTH] = {}, w[WIDTH] = {};
  for (int i = 0; i < WIDTH; i++) v[i] = w[i] = mulmod(0x9E3779B97F4A7C15ull % gl::P, (u64)(i + 1 + 13 * salt));
  rounds_ref(w, HEAD_ROUND, 3, true);
  for (int b = 0; b < BLOCKS; b++) rounds_ref(w, FIRST_BLOCK + 3 * b, 3, false);
  rounds_ref(w, TAIL_ROUND, 2, false);
  block_emulated(v, TBL.kh, true);
  for (int b = 0; b < BLOCKS; b++) block_emulated(v, TBL.kc[b], false);
  tail_emulated(v);
  for (int r = 0; r < WIDTH; r++)
    if (v[r] != w[r]) return false;
  return true;
}
constexpr bool tables_consistent() { return chain_consistent(0) && chain_consistent(7) && chain_consistent(101); }
static_assert(tables_consistent(), "poseidon_p3r.h: block tables disagree with the round-by-round definition");

typedef const Tables __attribute__((address_space(4))) * tbl_ptr;
typedef const u32 __attribute__((address_space(4))) * k_ptr;

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

// One row: sum_j coef[j] v_j + sum_{e in [E0, E0 + NEXTRA)} coef[12 + e] d_e + k, on 32-bit halves.
template <int E0, int NEXTRA, bool WITH_K = true>
__device__ __forceinline__ u64 row(const u32* lo, const u32* hi, const Row16& cf, const u32* dlo, const u32* dhi,
                                   u32 klo, u32 khi) {
  u64 al = mul_s(lo[0], cf.c[0]), ah = mul_s(hi[0], cf.c[0]);
#pragma unroll
  for (int j = 1; j < WIDTH; j++) {
    mad_s(al, lo[j], cf.c[j]);
    mad_s(ah, hi[j], cf.c[j]);
  }
#pragma unroll
  for (int e = E0; e < E0 + NEXTRA; e++) {
    mad_s(al, dlo[e], cf.c[12 + e]);
    mad_s(ah, dhi[e], cf.c[12 + e]);
  }
  if (WITH_K) {
    add_s(al, klo);
    add_s(ah, khi);
  }
  return reduce_row(al, ah);
}
// Row 0 of M (inline-constant entries 25, 15, 41, ..., 20) applied to the state, + 25 d (if WITH_D) + k
template <bool WITH_D>
__device__ __forceinline__ u64 row0_m(const u32* lo, const u32* hi, u32 dlo, u32 dhi, u32 klo, u32 khi) {
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
  if (WITH_D) {
    mad_k<25>(al, dlo);
    mad_k<25>(ah, dhi);
  }
  add_s(al, klo);
  add_s(ah, khi);
  return reduce_row(al, ah);
}
__device__ __forceinline__ void load_row(Row16& cf, const Row16 __attribute__((address_space(4))) * src) {
#pragma unroll
  for (int j = 0; j < 16; j++) cf.c[j] = src->c[j];
}
// d = sbox(x) - x for ANY u64 x (the row reduction's output as it is): the lazy subtraction of gl_lazy.h, both borrows
// corrected, 6 VALU -- the canonical form (gl::canon of x, then gl::sub) cost 10 per partial round.
__device__ __forceinline__ void delta(u64 x, u32& dlo, u32& dhi) {
  const u64 d = gl::sub_nc(sbox(x), x);
  dlo = (u32)d;
  dhi = (u32)(d >> 32);
}

// Three linear layers with the S-boxes of lane 0 in between.
//   HEAD = false: s = state entering a partial round (constants added); on return the state three rounds later.
//   HEAD = true : s = S-box outputs of the last leading full round; the first layer is that round's MDS.
template <bool HEAD>
__device__ __forceinline__ void three_rounds(u64 s[WIDTH], tbl_ptr tp, k_ptr kc) {
  u32 lo[WIDTH], hi[WIDTH];
#pragma unroll
  for (int i = 0; i < WIDTH; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  u32 dlo[3] = {0, 0, 0}, dhi[3] = {0, 0, 0};
  if (!HEAD) delta(s[0], dlo[0], dhi[0]);
  u64 x = row0_m<!HEAD>(lo, hi, dlo[0], dhi[0], kc[0], kc[1]);
  delta(x, dlo[1], dhi[1]);
  {
    Row16 cf;
    load_row(cf, &tp->r2);
    x = row<HEAD ? 1 : 0, HEAD ? 1 : 2>(lo, hi, cf, dlo, dhi, kc[2], kc[3]);
  }
  delta(x, dlo[2], dhi[2]);
#pragma unroll
  for (int r = 0; r < WIDTH; r++) {
    Row16 cf;
    load_row(cf, &tp->t3[r]);
    if (r == 0)
      s[r] = row<HEAD ? 1 : 0, HEAD ? 2 : 3, true>(lo, hi, cf, dlo, dhi, kc[4], kc[5]);
    else   // constants of words 1..11 are deferred to the tail block
      s[r] = row<HEAD ? 1 : 0, HEAD ? 2 : 3, false>(lo, hi, cf, dlo, dhi, 0, 0);
  }
}
// The last two partial rounds.
__device__ __forceinline__ void two_rounds(u64 s[WIDTH], tbl_ptr tp) {
  u32 lo[WIDTH], hi[WIDTH];
#pragma unroll
  for (int i = 0; i < WIDTH; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
  const k_ptr kc = tp->kt;
  u32 dlo[2], dhi[2];
  delta(s[0], dlo[0], dhi[0]);
  const u64 x = row0_m<true>(lo, hi, dlo[0], dhi[0], kc[0], kc[1]);
  delta(x, dlo[1], dhi[1]);
#pragma unroll
  for (int r = 0; r < WIDTH; r++) {
    Row16 cf;
    load_row(cf, &tp->t2[r]);
    s[r] = row<0, 2>(lo, hi, cf, dlo, dhi, kc[4 + 2 * r], kc[5 + 2 * r]);
  }
}

}  // namespace p3r
