// Poseidon (v1) for a WHOLE WAVE of 64 states with the MDS layers of the full rounds on the matrix cores
// (device only; same permutation, same bits as poseidon.h -- upstream plonky2 @ 3de92d9 hash/poseidon.rs
// `mds_layer`, called from /root/reference/src/p3/mod.rs:260 through PoseidonHash).
//
// Why: a full round is 12 S-boxes (672 VALU) + an MDS layer of 24 v_mad_u64_u32 + 5 per output word (348 VALU),
// and the chip is VALU-issue bound on exactly this.  The MDS layer is a 12x12 matrix with
// 6-bit entries applied to 64-bit words: on the bytes of the state it is an integer GEMM.
// v_mfma_i32_32x32x32_i8 computes, per instruction,  D[32 x 32] += A[32 x 32] B[32 x 32]  in the otherwise idle
// matrix pipe and costs the VALU stream ~0-3 issue cycles when >= 24 VALU instructions separate two of them
// (profiles/r03_mfmabench.txt), so the layer is restated as
//     D[(i, b')][state] = sum_{(j, b)} ( M[i][j] [b = b'] ) * byte_b(s_j)            (K = (word, byte) = 96)
// i.e. A = the MDS matrix expanded by a byte delta (a constant), B = the state bytes THEMSELVES:
//   * "pair layout": state h of a batch of 32 lives in the lane pair (h, h + 32), lane h holding words
//     {0,1,4,5,8,9} and lane h + 32 words {2,3,6,7,10,11}; a wave holds two batches (A: states 0..31, B: 32..63)
//     in 6 + 6 64-bit registers -- the same 24 VGPRs as one state per lane.  v_permlane32_swap of register w with
//     register w + 2 converts between the two layouts (12 swaps).  In pair layout the two words (16 bytes) a lane
//     holds for K tile t ARE its fragment of the B operand (lane = column = state, lane half = K half), so the
//     operand needs no data movement at all, only `xor 0x80808080` (the instruction multiplies signed bytes);
//   * rows are ordered so that lane half h' receives, in accumulator registers 8 s .. 8 s + 7, the eight byte sums
//     of output word 4 u + 2 h' + s -- exactly the words that lane holds -- with (D_k, D_{k+4}) in adjacent
//     registers: as 64-bit pairs V_k = D_k + 2^32 D_{k+4} the word is V_0 + 2^8 V_1 + 2^16 V_2 + 2^24 V_3,
//     recombined and reduced mod p in 11 VALU (instead of 29);
//   * a fourth K tile multiplies a constant operand (bytes 127 | 1) and injects, per row, the +128 * rowsum bias of
//     the signed bytes plus the matching byte of the NEXT round's constant (every partial sum stays >= 0 and
//     < 2^17, and no lane-dependent constant is needed on the VALU side);
//   * the circulant structure leaves 3 distinct state tiles (tile (u, t) depends on (t - u) mod 3) + one with the
//     diagonal entry: 16 VGPRs resident.
// The 24 MFMAs of a layer are issued one BEFORE each half S-box of the OTHER batch (the batches run half a round
// apart), so they never queue behind each other and the accumulators are read >= 28 VALU instructions after the last
// write (the MFMA -> VALU wait states are met by construction: inline asm is opaque to the compiler's hazard
// recogniser, tools/mdsbench.hip shows what happens otherwise).
// The partial rounds keep the M^3 blocks of poseidon_p3r.h in one-state-per-lane layout.
// Exact integer arithmetic throughout: bit-identical to permute_dev (tests/test_gpu_primitives.py, tools/mdsbench).
#pragma once
#include "poseidon.h"

#if defined(__HIP_DEVICE_COMPILE__)
namespace poseidon {
namespace mx {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int N_LAYERS = 7;  // the MDS layers of rounds 0, 1, 2 and 26 .. 29 (round 3's opens the head block of p3r)
constexpr int layer_round(int l) { return l < 3 ? l : HALF_FULL + N_PARTIAL + (l - 3); }

struct alignas(16) Tables {   // read as 16-byte vectors
  u32 st[4][64][4];             // state tiles: kind (t - u) mod 3, and kind 3 = tile (0, 0) with the diagonal entry
  u32 ct[N_LAYERS][3][64][4];   // constant tiles per layer and row tile u
};
constexpr u32 mds_entry(int i, int w) { return MDS_CIRC[(w - i + WIDTH) % WIDTH] + ((i == 0 && w == 0) ? MDS_DIAG0 : 0u); }
// lane L = (m = L & 31, h = L >> 5): row m of the A tile, K half h.  Row m is accumulator register
// r = (m & 3) + 4 (m >> 3) of lane half h' = (m >> 2) & 1 (C/D layout of the 32x32 shapes), which is made byte
// b' = ((r & 7) >> 1) + 4 (r & 1) of output word 4 u + 2 h' + (r >> 3).
constexpr void row_of_lane(int L, int u, int& i, int& bp) {
  const int m = L & 31, r = (m & 3) + 4 * (m >> 3), hp = (m >> 2) & 1;
  i = 4 * u + 2 * hp + (r >> 3);
  bp = ((r & 7) >> 1) + 4 * (r & 1);
}
constexpr void pack16(u32* o, const unsigned char* by) {
  for (int d = 0; d < 4; d++) o[d] = (u32)by[4 * d] | ((u32)by[4 * d + 1] << 8) | ((u32)by[4 * d + 2] << 16) | ((u32)by[4 * d + 3] << 24);
}
constexpr Tables make_tables() {
  Tables T{};
  for (int kind = 0; kind < 4; kind++)
    for (int L = 0; L < 64; L++) {
      const int u = 0, t = kind == 3 ? 0 : kind, h = L >> 5;   // tile (0, kind); kind 0 without the diagonal
      int i = 0, bp = 0;
      row_of_lane(L, u, i, bp);
      unsigned char by[16] = {};
      for (int s = 0; s < 2; s++) {
        const int w = 4 * t + 2 * h + s;
        u32 e = MDS_CIRC[(w - i + WIDTH) % WIDTH];
        if (kind == 3 && i == 0 && w == 0) e += MDS_DIAG0;
        by[8 * s + bp] = (unsigned char)e;
      }
      pack16(T.st[kind][L], by);
    }
  for (int l = 0; l < N_LAYERS; l++)
    for (int u = 0; u < 3; u++)
      for (int L = 0; L < 64; L++) {
        const int h = L >> 5;
        int i = 0, bp = 0;
        row_of_lane(L, u, i, bp);
        const int R = layer_round(l);
        const u64 rc = R + 1 < N_ROUNDS ? RC[WIDTH * (R + 1) + i] : 0;
        u32 rowsum = 0;
        for (int w = 0; w < WIDTH; w++) rowsum += mds_entry(i, w);
        u32 tot = 128 * rowsum + (u32)((rc >> (8 * bp)) & 255), X = tot / 127, Y = tot % 127;
        unsigned char by[16] = {};
        if (h == 0) {
          int k = 0;
          while (X) {
            const u32 v = X < 127 ? X : 127;
            by[k++] = (unsigned char)v;
            X -= v;
          }
        } else {
          by[0] = (unsigned char)Y;
        }
        pack16(T.ct[l][u][L], by);
      }
  return T;
}
static constexpr Tables TBL = make_tables();
constexpr int tile_kind(int u, int t) { return (u == 0 && t == 0) ? 3 : (t - u + 3) % 3; }
// the state tiles really depend on (t - u) mod 3 only (and the diagonal sits in tile (0, 0))
constexpr bool tiles_consistent() {
  for (int u = 0; u < 3; u++)
    for (int t = 0; t < 3; t++)
      for (int L = 0; L < 64; L++) {
        int i = 0, bp = 0;
        row_of_lane(L, u, i, bp);
        unsigned char by[16] = {};
        for (int s = 0; s < 2; s++) by[8 * s + bp] = (unsigned char)mds_entry(i, 4 * t + 2 * (L >> 5) + s);
        u32 o[4] = {};
        pack16(o, by);
        for (int d = 0; d < 4; d++)
          if (o[d] != TBL.st[tile_kind(u, t)][L][d]) return false;
      }
  return true;
}
static_assert(tiles_consistent(), "poseidon_mfma.h: state tiles");

typedef const v4i __attribute__((address_space(4))) * tile_ptr;

struct Pow256 {
  u32 p1, p2, p3;  // 2^8, 2^16, 2^24 in SGPRs (no VOP3 literals on gfx9)
};
struct Ctx {
  v4i tk[4];   // the four state tiles of this lane
  v4i bc;      // B operand of the constant tile: bytes 127 (lanes 0..31) | 1 (lanes 32..63)
  tile_ptr ct; // constant tiles, this lane's column
  Pow256 pw;
};
__device__ __forceinline__ Ctx make_ctx(int lane) {
  Ctx c;
  tile_ptr st = (tile_ptr)&TBL.st[0][0][0];
#pragma unroll
  for (int k = 0; k < 4; k++) c.tk[k] = st[k * 64 + lane];
  const int cv = lane < 32 ? 0x7f7f7f7f : 0x01010101;
  c.bc = v4i{cv, cv, cv, cv};
  c.ct = (tile_ptr)&TBL.ct[0][0][0][0] + lane;
  c.pw.p1 = 1u << 8;
  c.pw.p2 = 1u << 16;
  c.pw.p3 = 1u << 24;
  asm("" : "+s"(c.pw.p1), "+s"(c.pw.p2), "+s"(c.pw.p3));
  return c;
}

__device__ __forceinline__ void swap_dw(u32& a, u32& b) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
// lane layout <-> pair layout for word pair (w, w + 2): register w becomes batch A's, register w + 2 batch B's
__device__ __forceinline__ void swap_words(u64& x, u64& y) {
  u32 a0 = (u32)x, a1 = (u32)(x >> 32), b0 = (u32)y, b1 = (u32)(y >> 32);
  swap_dw(a0, b0);
  swap_dw(a1, b1);
  x = gl::make64(a0, a1);
  y = gl::make64(b0, b1);
}
// s[12] (one state per lane) -> A[2t + sl] = register of word 4t + sl, B[2t + sl] = register of word 4t + 2 + sl
__device__ __forceinline__ void to_pairs(const u64 s[WIDTH], u64 A[6], u64 B[6]) {
  // the last writer of s may sit inside an asm block: the VALU-write -> v_permlane-read wait states by hand
  asm volatile("s_nop 1");
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
      A[2 * t + sl] = s[4 * t + sl];
      B[2 * t + sl] = s[4 * t + 2 + sl];
      swap_words(A[2 * t + sl], B[2 * t + sl]);
    }
}
__device__ __forceinline__ void from_pairs(u64 s[WIDTH], u64 A[6], u64 B[6], u32 rows = 0xFFF) {
  asm volatile("s_nop 1");
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const bool want = (rows >> (4 * t)) & 0xFu;  // wave-uniform: a word group the caller will read
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
      if (want) swap_words(A[2 * t + sl], B[2 * t + sl]);
      // assigned either way (unspecified values for a group not wanted), so that the caller's old words are dead
      s[4 * t + sl] = A[2 * t + sl];
      s[4 * t + 2 + sl] = B[2 * t + sl];
    }
  }
}

// 8 byte sums of one word, (D_k, D_{k+4}) in adjacent registers, each in [0, 2^17) -> the word mod p (non-canonical).
//   word = AL + 2^32 AH,  AL = (D_0 + 2^32 D_4) + 2^8 D_1 + 2^16 D_2 + 2^24 D_3 < 2^50  (the pair (D_0, D_4) is the
//   64-bit addend of the first multiply-add),  AH = 2^8 D_5 + 2^16 D_6 + 2^24 D_7 < 2^42.
// (v_lshl_add_u64 shifts by at most 4, so the byte weights go through v_mad_u64_u32 with SGPR multipliers.)
__device__ __forceinline__ u64 recombine(int d0, int d1, int d2, int d3, int d4, int d5, int d6, int d7, const Pow256& k) {
  // register order: d0 = D_0, d1 = D_4, d2 = D_1, d3 = D_5, d4 = D_2, d5 = D_6, d6 = D_3, d7 = D_7
  u64 AL = gl::make64((u32)d0, (u32)d1), AH, dm;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(AL), "=s"(dm) : "v"(d2), "s"(k.p1));
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(AH), "=s"(dm) : "v"(d3), "s"(k.p1));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(AL), "=s"(dm) : "v"(d4), "s"(k.p2));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(AH), "=s"(dm) : "v"(d5), "s"(k.p2));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(AL), "=s"(dm) : "v"(d6), "s"(k.p3));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(AH), "=s"(dm) : "v"(d7), "s"(k.p3));
  return p3r::reduce_row(AL, AH);
}

__device__ __forceinline__ void xor_operand(const u64 X[6], v4i b[3]) {
#pragma unroll
  for (int t = 0; t < 3; t++) {
    b[t][0] = (int)((u32)X[2 * t] ^ 0x80808080u);
    b[t][1] = (int)((u32)(X[2 * t] >> 32) ^ 0x80808080u);
    b[t][2] = (int)((u32)X[2 * t + 1] ^ 0x80808080u);
    b[t][3] = (int)((u32)(X[2 * t + 1] >> 32) ^ 0x80808080u);
  }
}
// MFMA number k of a layer (k = 3 t + u: consecutive ones are independent); umask: row tiles wanted
template <int K>
__device__ __forceinline__ void mfma_step(v16i acc[3], const v4i b[3], const Ctx& c, const v4i ctile[3], u32 umask) {
  constexpr int t = K / 3, u = K % 3;
  if (!((umask >> u) & 1u)) return;
  const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (t == 0)
    acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(c.tk[tile_kind(u, 0)], b[0], zero, 0, 0, 0);
  else if constexpr (t < 3)
    acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(c.tk[tile_kind(u, t)], b[t], acc[u], 0, 0, 0);
  else
    acc[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ctile[u], c.bc, acc[u], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void load_ctiles(v4i ctile[3], const Ctx& c, int layer) {
#pragma unroll
  for (int u = 0; u < 3; u++) ctile[u] = c.ct[(layer * 3 + u) * 64];
}
__device__ __forceinline__ void recombine6(u64 X[6], const v16i acc[3], const Ctx& c, u32 umask = 7) {
#pragma unroll
  for (int u = 0; u < 3; u++) {
    if (!((umask >> u) & 1u)) continue;
    X[2 * u] = recombine(acc[u][0], acc[u][1], acc[u][2], acc[u][3], acc[u][4], acc[u][5], acc[u][6], acc[u][7], c.pw);
    X[2 * u + 1] = recombine(acc[u][8], acc[u][9], acc[u][10], acc[u][11], acc[u][12], acc[u][13], acc[u][14], acc[u][15], c.pw);
  }
}
__device__ __forceinline__ void sbox6(u64 X[6]) {
#pragma unroll
  for (int i = 0; i < 6; i++) X[i] = sbox(X[i]);
}
__device__ __forceinline__ void mfma12(v16i acc[3], const v4i b[3], const Ctx& c, const v4i ctile[3], u32 umask) {
  mfma_step<0>(acc, b, c, ctile, umask);  mfma_step<1>(acc, b, c, ctile, umask);  mfma_step<2>(acc, b, c, ctile, umask);
  mfma_step<3>(acc, b, c, ctile, umask);  mfma_step<4>(acc, b, c, ctile, umask);  mfma_step<5>(acc, b, c, ctile, umask);
  mfma_step<6>(acc, b, c, ctile, umask);  mfma_step<7>(acc, b, c, ctile, umask);  mfma_step<8>(acc, b, c, ctile, umask);
  mfma_step<9>(acc, b, c, ctile, umask);  mfma_step<10>(acc, b, c, ctile, umask); mfma_step<11>(acc, b, c, ctile, umask);
  // nothing of this wave to put between the last write and the first read: wait it out (tied to the accumulators,
  // so that their readers cannot be scheduled above it)
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
}

// The six S-boxes of one batch with the twelve MFMAs of the other batch's layer between their halves.
__device__ __forceinline__ void sbox6_mfma12(u64 X[6], v16i acc[3], const v4i b[3], const Ctx& c, const v4i ctile[3],
                                             u32 umask) {
#if defined(P25_MX_SERIAL)
  sbox6(X);
  asm volatile("s_nop 7");
  mfma12(acc, b, c, ctile, umask);
  return;
#endif
#define P25_SB_HALF(i, K0, K1)                 \
  {                                            \
    mfma_step<K0>(acc, b, c, ctile, umask);    \
    const u64 x = X[i];                        \
    const u64 x2 = gl::mul_nc(x, x);           \
    const u64 x4 = gl::mul_nc(x2, x2);         \
    __builtin_amdgcn_sched_barrier(0);         \
    mfma_step<K1>(acc, b, c, ctile, umask);    \
    const u64 x3 = gl::mul_nc(x, x2);          \
    X[i] = gl::mul_nc(x3, x4);                 \
    __builtin_amdgcn_sched_barrier(0);         \
  }
  P25_SB_HALF(0, 0, 1)
  P25_SB_HALF(1, 2, 3)
  P25_SB_HALF(2, 4, 5)
  P25_SB_HALF(3, 6, 7)
  P25_SB_HALF(4, 8, 9)
  P25_SB_HALF(5, 10, 11)
#undef P25_SB_HALF
}
// N full rounds: A, B = the S-box INPUTS of the first one (pair layout); layers L0 .. L0 + N - 1.
//   SBOX_AFTER: one more S-box layer after the last MDS (round 3, whose MDS opens the head block of the partial rounds).
//   umask_last: row tiles of the LAST layer the caller needs (wave-uniform).
template <int L0, int N, bool SBOX_AFTER>
__device__ __forceinline__ void full_rounds(u64 A[6], u64 B[6], const Ctx& c, u32 umask_last) {
  v16i acc[3];
  v4i b[3], ctile[3];
  sbox6(A);
#pragma unroll
  for (int l = 0; l < N; l++) {
    const bool last = l == N - 1;
    load_ctiles(ctile, c, L0 + l);
    // the last layer issues every MFMA as well (they cost next to nothing; branches around them do) and only
    // recombines the row tiles the caller reads
    const u32 umask = (last && !SBOX_AFTER) ? umask_last : 7u;
    xor_operand(A, b);
    sbox6_mfma12(B, acc, b, c, ctile, 7);       // B's S-boxes of this round under A's layer
    recombine6(A, acc, c, umask);
    xor_operand(B, b);
    if (!last || SBOX_AFTER) {
      sbox6_mfma12(A, acc, b, c, ctile, 7);     // A's S-boxes of the NEXT round under B's layer
      recombine6(B, acc, c);
    } else {
      mfma12(acc, b, c, ctile, 7);
      recombine6(B, acc, c, umask);
    }
  }
  if (SBOX_AFTER) sbox6(B);
}

// The permutation of permute_dev for the 64 states of a wave (every lane active, one state per lane on entry and
// exit).  rows: the output words the caller reads (ROWS_DIGEST / ROWS_CAPACITY / ROWS_ALL), wave-uniform.
__device__ __forceinline__ void permute_wave(u64 s[WIDTH], u32 rows, const Ctx& c) {
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = add_rc(s[i], RC[i]);
  u64 A[6], B[6];
  to_pairs(s, A, B);
  full_rounds<0, 3, true>(A, B, c, 7);
  from_pairs(s, A, B);
  p3r::tbl_ptr tp = (p3r::tbl_ptr)&p3r::TBL;
  asm("" : "+s"(tp));
  p3r::three_rounds<true>(s, tp, tp->kh);
  for (int blk = 0; blk < p3r::BLOCKS; blk++) p3r::three_rounds<false>(s, tp, tp->kc[blk]);
  p3r::two_rounds(s, tp);
  to_pairs(s, A, B);
  const u32 umask = ((rows & 0x00F) ? 1u : 0u) | ((rows & 0x0F0) ? 2u : 0u) | ((rows & 0xF00) ? 4u : 0u);
  full_rounds<3, 4, false>(A, B, c, umask);
  from_pairs(s, A, B, rows);
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(s[i]);
}

}  // namespace mx
}  // namespace poseidon
#endif
