// Poseidon (v1) permutation over Goldilocks, width 12, as used by plonky2's `PoseidonHash`
// (the prover's own hash: `type C = PoseidonGoldilocksConfig`, /root/reference/src/p3/mod.rs:229).
// The algorithm lives in the absent upstream crate plonky2 @ 3de92d9 (hash/poseidon.rs,
// hash/poseidon_goldilocks.rs); it is restated here from its published definition:
//   8 full rounds (4 + 4) and 22 partial rounds, S-box x^7, MDS = circulant(MDS_CIRC) + diag(MDS_DIAG),
//   360 round constants (tools/gen_poseidon_constants.py), every round = add constants, S-box, MDS.
// Pinned by the four known-answer vectors in the reference tree at
// src/common/poseidon2/poseidon2_goldilocks.rs:190-211 (tests/test_oracle_kats.py, tests/test_gpu_*).
//
// Sponge conventions (upstream hashing.rs): overwrite mode, rate 8, capacity 4, 4-word digest;
// `two_to_one(l, r)` = permute(l || r || 0000)[0..4]; `hash_or_noop` pads inputs of <= 4 words.
#pragma once
#include "gl_lazy.h"

namespace poseidon {

constexpr int WIDTH = 12;
constexpr int RATE = 8;
constexpr int HALF_FULL = 4;
constexpr int N_PARTIAL = 22;
constexpr int N_ROUNDS = 30;

static constexpr u64 RC[360] = {
#include "poseidon_constants.inc"
};
static constexpr u32 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static constexpr u32 MDS_DIAG0 = 8;  // MDS_DIAG = [8, 0, ..., 0]

GL_HD u64 sbox(u64 x) {
  u64 x2 = gl::mul_nc(x, x);
  u64 x4 = gl::mul_nc(x2, x2);
  u64 x3 = gl::mul_nc(x, x2);
  return gl::mul_nc(x3, x4);
}

// s[] any u64 representatives in, non-canonical out.  out[r] = sum_i s[(i+r)%12]*CIRC[i] + s[r]*DIAG[r].
GL_HD void mds(u64 s[WIDTH]) {
  u64 lo[WIDTH], hi[WIDTH];
#pragma unroll
  for (int i = 0; i < WIDTH; i++) {
    lo[i] = s[i] & gl::EPS;
    hi[i] = s[i] >> 32;
  }
#pragma unroll
  for (int r = 0; r < WIDTH; r++) {
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < WIDTH; i++) {
      al += lo[(i + r) % WIDTH] * MDS_CIRC[i];
      ah += hi[(i + r) % WIDTH] * MDS_CIRC[i];
    }
    if (r == 0) {
      al += lo[0] * MDS_DIAG0;
      ah += hi[0] * MDS_DIAG0;
    }
    // value = al + ah * 2^32  (al, ah < 2^42)
    u64 l64 = al + (ah << 32);
    u32 h32 = (u32)(ah >> 32) + (l64 < al ? 1u : 0u);
    s[r] = gl::reduce96(l64, h32);
  }
}

// add round constants: s any u64, rc canonical -> result any u64 (mod-p correct)
GL_HD u64 add_rc(u64 x, u64 c) {
  u64 s = x + c;
  return s < x ? s + gl::EPS : s;  // on wrap: +2^64 == +EPS; cannot wrap twice since c < p
}

// The permutation in its defining (naive) form with the S-box inputs handed to `tr(k, value)` (canonical), in the
// order upstream's PoseidonGate stores them as wires: full rounds 1..3 (36), partial rounds (22, lane 0), full
// rounds 26..29 (48).  Used by the PoseidonGate witness generator and constraint evaluator (recursion, SURVEY
// 8f-4) -- not by the hashing kernels, which run the fused form below.
template <class Tracer>
GL_HD void permute_naive_trace(u64 s[WIDTH], Tracer&& tr) {
  int k = 0;
  for (int r = 0; r < N_ROUNDS; r++) {
    for (int i = 0; i < WIDTH; i++) s[i] = add_rc(s[i], RC[WIDTH * r + i]);
    if (r < HALF_FULL || r >= HALF_FULL + N_PARTIAL) {
      if (r != 0)
        for (int i = 0; i < WIDTH; i++) {
          s[i] = gl::canon(s[i]);
          tr(k++, s[i]);
        }
      for (int i = 0; i < WIDTH; i++) s[i] = sbox(s[i]);
    } else {
      s[0] = gl::canon(s[0]);
      tr(k++, s[0]);
      s[0] = sbox(s[0]);
    }
    mds(s);
  }
  for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- gfx950 device form -------------------------------------------------------------------------
// Round constants split into zero-extended 32-bit halves, one (lo, hi) pair per constant, plus a
// zero row: the constants of round r+1 are folded into round r's MDS accumulators (they ride in as
// the 64-bit addend of the first v_mad_u64_u32 of each row, read straight from SGPRs), so only
// round 0 pays for a separate modular addition.
struct RcSplit {
  u64 v[(N_ROUNDS + 1) * WIDTH * 2];
};
constexpr RcSplit make_rc_split() {
  RcSplit t{};
  for (int i = 0; i < N_ROUNDS * WIDTH; i++) {
    t.v[2 * i] = RC[i] & 0xFFFFFFFFull;
    t.v[2 * i + 1] = RC[i] >> 32;
  }
  return t;
}
static constexpr RcSplit RC_SPLIT = make_rc_split();

template <int C>
__device__ __forceinline__ void mad_k(u64& acc, u32 x) {
  u64 dm;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(dm) : "v"(x), "n"(C));
}
// s <- MDS * s + k, k = the (lo, hi) pairs of the next round's constants.  288 v_mad_u64_u32 with inline
// constants + 5 instructions per row for the reduction (the compiler's version of mds(): shifts for the
// power-of-two entries through v_mov'd pairs, ~23 non-mad instructions per row).
typedef const u64 __attribute__((address_space(4))) * rc_ptr;  // constant address space: scalar loads
// `rows`: bit r set = output word r is wanted (wave-uniform; the last layer of a sponge permutation only
// needs the capacity words, or the digest words -- see permute_rows).
__device__ __forceinline__ void mds_rc(u64 s[WIDTH], rc_ptr k, u32 rows = 0xFFF) {
  u32 lo[WIDTH], hi[WIDTH];
#pragma unroll
  for (int i = 0; i < WIDTH; i++) {
    lo[i] = (u32)s[i];
    hi[i] = (u32)(s[i] >> 32);
  }
#pragma unroll
  for (int r = 0; r < WIDTH; r++) {
    if (!((rows >> r) & 1u)) continue;
    u64 al, ah, dm, t;
    asm("v_mad_u64_u32 %0, %1, %2, 17, %3" : "=v"(al), "=s"(dm) : "v"(lo[r]), "s"(k[2 * r]));
    asm("v_mad_u64_u32 %0, %1, %2, 17, %3" : "=v"(ah), "=s"(dm) : "v"(hi[r]), "s"(k[2 * r + 1]));
#define P25_MDS_TERM(i, C)            \
  mad_k<C>(al, lo[(i + r) % WIDTH]); \
  mad_k<C>(ah, hi[(i + r) % WIDTH]);
    P25_MDS_TERM(1, 15)
    P25_MDS_TERM(2, 41)
    P25_MDS_TERM(3, 16)
    P25_MDS_TERM(4, 2)
    P25_MDS_TERM(5, 28)
    P25_MDS_TERM(6, 13)
    P25_MDS_TERM(7, 13)
    P25_MDS_TERM(8, 39)
    P25_MDS_TERM(9, 18)
    P25_MDS_TERM(10, 34)
    P25_MDS_TERM(11, 20)
#undef P25_MDS_TERM
    if (r == 0) {
      mad_k<8>(al, lo[0]);
      mad_k<8>(ah, hi[0]);
    }
    // value = al + ah * 2^32, al < 2^43, ah < 2^42:  X = al + (ah >> 32) * (2^32 - 1) < 2^44, then
    // add (ah & 0xffffffff) << 32; a carry out of bit 64 is worth 2^32 - 1 and cannot ripple
    // (after a wrap the high word is < 2^12).
    u32 ahl = (u32)ah, ahh = (u32)(ah >> 32);
    u64 X;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(X), "=s"(dm) : "v"(ahh), "v"(al));
    u32 x0 = (u32)X, x1 = (u32)(X >> 32);
    asm("v_add_co_u32_e32 %1, vcc, %1, %3\n\ts_nop 1\n\t"
        "v_subbrev_co_u32_e64 %0, %2, 0, %0, vcc\n\t"
        "s_andn2_b64 %2, vcc, %2\n\t"
        "v_addc_co_u32_e64 %1, %2, 0, %1, %2"
        : "+v"(x0), "+v"(x1), "=&s"(t)
        : "v"(ahl)
        : "vcc", "scc");  // s_andn2 writes SCC
    s[r] = gl::make64(x0, x1);
  }
}
static_assert(MDS_CIRC[0] == 17 && MDS_CIRC[1] == 15 && MDS_CIRC[2] == 41 && MDS_CIRC[3] == 16 && MDS_CIRC[4] == 2 &&
                  MDS_CIRC[5] == 28 && MDS_CIRC[6] == 13 && MDS_CIRC[7] == 13 && MDS_CIRC[8] == 39 &&
                  MDS_CIRC[9] == 18 && MDS_CIRC[10] == 34 && MDS_CIRC[11] == 20 && MDS_DIAG0 == 8,
              "mds_rc hard-codes the MDS entries as inline constants");

#include "poseidon_p3r.h"

// x^7 of the round-0 constants of the capacity words: what the first S-box layer yields for them when the capacity
// enters as zero (the first permutation of every sponge, every two_to_one) -- 4 of that layer's 12 S-boxes.
constexpr u64 sbox_const(u64 x) {
  const unsigned __int128 p = gl::P;
  u64 x2 = (u64)((unsigned __int128)x * x % p), x4 = (u64)((unsigned __int128)x2 * x2 % p), x3 = (u64)((unsigned __int128)x * x2 % p);
  return (u64)((unsigned __int128)x3 * x4 % p);
}
static constexpr u64 SBOX_RC0_CAP[4] = {sbox_const(RC[8]), sbox_const(RC[9]), sbox_const(RC[10]), sbox_const(RC[11])};

// cap0 (wave-uniform): the caller guarantees s[8..11] == 0
__device__ __forceinline__ void permute_dev(u64 s[WIDTH], u32 rows, bool cap0 = false) {
#pragma unroll
  for (int i = 0; i < RATE; i++) s[i] = add_rc(s[i], RC[i]);
  // Opaque base pointer: otherwise every one of the 24 scalar loads per round recomputes the table's
  // pc-relative address (s_getpc + 4 scalar adds each).
  rc_ptr rc = (rc_ptr)RC_SPLIT.v;
  asm("" : "+s"(rc));
  int r = 1;
  // full round 0: the capacity words' S-boxes are constants when the capacity is zero
#pragma unroll
  for (int i = 0; i < RATE; i++) s[i] = sbox(s[i]);
  if (cap0) {
#pragma unroll
    for (int i = RATE; i < WIDTH; i++) s[i] = SBOX_RC0_CAP[i - RATE];
  } else {
#pragma unroll
    for (int i = RATE; i < WIDTH; i++) s[i] = sbox(add_rc(s[i], RC[i]));
  }
  mds_rc(s, rc + 2 * WIDTH * r);
  r++;
  // full rounds 1, 2 with their MDS layers; round 3's S-boxes only -- its MDS is the first layer of the head block
  for (int k = 1; k < HALF_FULL - 1; k++, r++) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = sbox(s[i]);
    mds_rc(s, rc + 2 * WIDTH * r);
  }
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = sbox(s[i]);
  // the 22 partial rounds: head block (MDS of round 3 + rounds 4, 5), six blocks of three, tail block of two
  p3r::tbl_ptr tp = (p3r::tbl_ptr)&p3r::TBL;
  asm("" : "+s"(tp));
  p3r::three_rounds<true>(s, tp, tp->kh);
  for (int b = 0; b < p3r::BLOCKS; b++) p3r::three_rounds<false>(s, tp, tp->kc[b]);
  p3r::two_rounds(s, tp);
  r = HALF_FULL + N_PARTIAL + 1;
  for (int k = 0; k < HALF_FULL - 1; k++, r++) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = sbox(s[i]);
    mds_rc(s, rc + 2 * WIDTH * r);
  }
  // last round: the zero constant row, and only the output words the caller will read
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = sbox(s[i]);
  mds_rc(s, rc + 2 * WIDTH * N_ROUNDS, rows);
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}
#endif

// Canonical in, canonical out.  permute_rows: only the output words whose bit is set in `rows` are
// defined afterwards (the others hold unspecified canonical values).  In an overwrite-mode sponge every
// absorbing permutation is followed by 8 fresh inputs, so only its capacity words 8..11 matter, and the
// last one only yields the 4 digest words: 8 of the 12 rows of the final MDS layer are never computed.
GL_HD void permute_rows(u64 s[WIDTH], u32 rows, bool cap0 = false) {
#if defined(__HIP_DEVICE_COMPILE__)
  permute_dev(s, rows, cap0);
#else
  (void)rows;
  (void)cap0;
  int r = 0;
  for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = sbox(add_rc(s[i], RC[12 * r + i]));
    mds(s);
  }
  // (The sparse-matrix "fast partial rounds" form -- tools/gen_poseidon_fast.py -- was measured on
  // MI355X and is slower: 128-bit carry chains cost more than the dense small-constant layer below.)
  for (int k = 0; k < N_PARTIAL; k++, r++) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = add_rc(s[i], RC[12 * r + i]);
    s[0] = sbox(s[0]);
    mds(s);
  }
  for (int k = 0; k < HALF_FULL; k++, r++) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = sbox(add_rc(s[i], RC[12 * r + i]));
    mds(s);
  }
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
#endif
}

GL_HD void permute(u64 s[WIDTH]) { permute_rows(s, 0xFFF); }
constexpr u32 ROWS_DIGEST = 0x00F, ROWS_CAPACITY = 0xF00, ROWS_ALL = 0xFFF;

// compress two 4-word digests (upstream `two_to_one`)
GL_HD void two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
  u64 s[WIDTH];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[i] = l[i];
    s[4 + i] = r[i];
    s[8 + i] = 0;
  }
  permute_rows(s, ROWS_DIGEST, true);
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = s[i];
}

// hash_no_pad over a strided sequence: element k is in[k * stride].
GL_HD void hash_no_pad_strided(const u64* in, size_t stride, int n, u64 out[4]) {
  u64 s[WIDTH];
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = 0;
  for (int off = 0; off < n; off += RATE) {
    int m = n - off < RATE ? n - off : RATE;
    for (int i = 0; i < m; i++) s[i] = in[(size_t)(off + i) * stride];
    const int next = n - (off + RATE);  // inputs still to absorb after this permutation
    permute_rows(s, next <= 0 ? ROWS_DIGEST : (next >= RATE ? ROWS_CAPACITY : ROWS_ALL), off == 0);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) out[i] = s[i];
}

// upstream `hash_or_noop`: inputs of <= 4 words are zero-padded instead of hashed.
GL_HD void hash_or_noop_strided(const u64* in, size_t stride, int n, u64 out[4]) {
  if (n <= 4) {
    for (int i = 0; i < 4; i++) out[i] = i < n ? in[(size_t)i * stride] : 0;
  } else {
    hash_no_pad_strided(in, stride, n, out);
  }
}

}  // namespace poseidon
