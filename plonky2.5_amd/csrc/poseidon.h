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
#include "gl.h"

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

// Canonical in, canonical out.
GL_HD void permute(u64 s[WIDTH]) {
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
}

// compress two 4-word digests (upstream `two_to_one`)
GL_HD void two_to_one(const u64 l[4], const u64 r[4], u64 out[4]) {
  u64 s[WIDTH];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    s[i] = l[i];
    s[4 + i] = r[i];
    s[8 + i] = 0;
  }
  permute(s);
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
    permute(s);
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
