// ORACLE -- test infrastructure only.  The TUNED CPU-baseline leg: the same Poseidon (v1)
// permutation, sponge and two_to_one as ref_hash.cpp -- upstream plonky2 @ 3de92d9 hash/poseidon.rs, hashing.rs,
// merkle_tree.rs, reached from /root/reference/src/p3/mod.rs:260 with `type C = PoseidonGoldilocksConfig` (:229) -- for EIGHT
// independent hashes at a time, one per 64-bit lane of an AVX-512 register (what upstream's own
// `arch/x86_64/avx512_goldilocks_field.rs` packing does for its CPU prover).  The defining round structure of
// ref_poseidon_naive (add constants, x^7, circulant MDS with entries <= 41), every value canonical after every operation,
// so the digests are bit-identical to the scalar oracle's; tests/test_oracle_kats.py checks that on the reference's four
// known-answer vectors and on random Merkle trees, and bench.py checks the whole proof's bytes.
// Only the Merkle-tree builders of ref_prover.cpp call this, and only after p25o_set_tuned(1): the checker (tests, smoke) runs
// the plain scalar code.
#include "ref_hash_x8.h"
#include <string.h>

static const u64 X8_RC[360] = {
#include "poseidon_constants.inc"
};
static const u32 X8_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};

bool ref_x8_available() { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }

#include "ref_field_x8.h"

X8 V v_pow7(V x) {
  V x2 = v_mul(x, x), x4 = v_mul(x2, x2), x3 = v_mul(x, x2);
  return v_mul(x3, x4);
}
// out[r] = sum_i s[(i + r) % 12] * CIRC[i] + 8 * s[0] (r = 0): on the 32-bit halves, sums < 2^42 (as ref_hash.cpp poseidon_mds)
__attribute__((target("avx512f,avx512dq"))) static void v_mds(V s[12]) {
  const V eps = v_eps();
  V lo[12], hi[12], o[12];
#pragma GCC unroll 12
  for (int i = 0; i < 12; i++) {
    lo[i] = _mm512_and_si512(s[i], eps);
    hi[i] = _mm512_srli_epi64(s[i], 32);
  }
  const V one = _mm512_set1_epi64(1);
#pragma GCC unroll 12
  for (int r = 0; r < 12; r++) {
    // two accumulators per half: the sums are short dependent chains
    V al0 = _mm512_mul_epu32(lo[r], _mm512_set1_epi64(X8_CIRC[0] + (r == 0 ? 8 : 0)));
    V ah0 = _mm512_mul_epu32(hi[r], _mm512_set1_epi64(X8_CIRC[0] + (r == 0 ? 8 : 0)));
    V al1 = _mm512_mul_epu32(lo[(r + 1) % 12], _mm512_set1_epi64(X8_CIRC[1]));
    V ah1 = _mm512_mul_epu32(hi[(r + 1) % 12], _mm512_set1_epi64(X8_CIRC[1]));
#pragma GCC unroll 5
    for (int i = 2; i < 12; i += 2) {
      al0 = _mm512_add_epi64(al0, _mm512_mul_epu32(lo[(i + r) % 12], _mm512_set1_epi64(X8_CIRC[i])));
      ah0 = _mm512_add_epi64(ah0, _mm512_mul_epu32(hi[(i + r) % 12], _mm512_set1_epi64(X8_CIRC[i])));
      al1 = _mm512_add_epi64(al1, _mm512_mul_epu32(lo[(i + 1 + r) % 12], _mm512_set1_epi64(X8_CIRC[i + 1])));
      ah1 = _mm512_add_epi64(ah1, _mm512_mul_epu32(hi[(i + 1 + r) % 12], _mm512_set1_epi64(X8_CIRC[i + 1])));
    }
    V al = _mm512_add_epi64(al0, al1), ah = _mm512_add_epi64(ah0, ah1);
    // al + ah * 2^32 as (hi64, lo64)
    V sh = _mm512_slli_epi64(ah, 32);
    V l64 = _mm512_add_epi64(al, sh);
    __mmask8 c = _mm512_cmplt_epu64_mask(l64, sh);
    V h64 = _mm512_srli_epi64(ah, 32);
    h64 = _mm512_mask_add_epi64(h64, c, h64, one);      // < 2^11
    V m = _mm512_mul_epu32(h64, eps);                   // h64 * 2^64 = h64 * (2^32 - 1)
    V x = _mm512_add_epi64(l64, m);
    __mmask8 c2 = _mm512_cmplt_epu64_mask(x, m);
    x = _mm512_mask_add_epi64(x, c2, x, eps);
    __mmask8 ge = _mm512_cmpge_epu64_mask(x, v_p());
    o[r] = _mm512_mask_sub_epi64(x, ge, x, v_p());
  }
#pragma GCC unroll 12
  for (int i = 0; i < 12; i++) s[i] = o[i];
}
// the permutation of ref_poseidon_naive on eight states (state word i of hash k in lane k of s[i])
__attribute__((target("avx512f,avx512dq"))) static void v_poseidon(V s[12]) {
  for (int r = 0; r < 30; r++) {
#pragma GCC unroll 12
    for (int i = 0; i < 12; i++) s[i] = v_add(s[i], _mm512_set1_epi64((long long)X8_RC[12 * r + i]));
    if (r < 4 || r >= 26) {
#pragma GCC unroll 12
      for (int i = 0; i < 12; i++) s[i] = v_pow7(s[i]);
    } else {
      s[0] = v_pow7(s[0]);
    }
    v_mds(s);
  }
}

__attribute__((target("avx512f,avx512dq"))) void ref_poseidon_x8(u64 states[8][12]) {
  V s[12];
  const V idx = _mm512_setr_epi64(0, 12, 24, 36, 48, 60, 72, 84);
  for (int i = 0; i < 12; i++) s[i] = _mm512_i64gather_epi64(idx, (const long long*)&states[0][i], 8);
  v_poseidon(s);
  for (int i = 0; i < 12; i++) _mm512_i64scatter_epi64((long long*)&states[0][i], idx, s[i], 8);
}

// hash_no_pad (overwrite-mode sponge, rate 8) of the eight consecutive rows leaves[(i0 + k) * width ...], width > 4
__attribute__((target("avx512f,avx512dq"))) void ref_hash_rows_x8(const u64* leaves, size_t width, size_t i0, RHash out[8]) {
  V s[12];
  for (int i = 0; i < 12; i++) s[i] = _mm512_setzero_si512();
  const long long w = (long long)width;
  const V idx = _mm512_setr_epi64(0, w, 2 * w, 3 * w, 4 * w, 5 * w, 6 * w, 7 * w);
  const u64* base = leaves + i0 * width;
  for (size_t off = 0; off < width; off += 8) {
    const size_t m = width - off < 8 ? width - off : 8;
    for (size_t i = 0; i < m; i++) s[i] = _mm512_i64gather_epi64(idx, (const long long*)(base + off + i), 8);
    v_poseidon(s);
  }
  const V oidx = _mm512_setr_epi64(0, 4, 8, 12, 16, 20, 24, 28);
  for (int i = 0; i < 4; i++) _mm512_i64scatter_epi64((long long*)&out[0].e[i], oidx, s[i], 8);
}

// parents[k] = two_to_one(children[2k], children[2k + 1]), k < 8 (children: 16 consecutive digests)
__attribute__((target("avx512f,avx512dq"))) void ref_two_to_one_x8(const RHash* children, RHash parents[8]) {
  V s[12];
  const V idx = _mm512_setr_epi64(0, 8, 16, 24, 32, 40, 48, 56);
  for (int i = 0; i < 8; i++) s[i] = _mm512_i64gather_epi64(idx, (const long long*)(&children[0].e[0] + i), 8);
  for (int i = 8; i < 12; i++) s[i] = _mm512_setzero_si512();
  v_poseidon(s);
  const V oidx = _mm512_setr_epi64(0, 4, 8, 12, 16, 20, 24, 28);
  for (int i = 0; i < 4; i++) _mm512_i64scatter_epi64((long long*)&parents[0].e[i], oidx, s[i], 8);
}
