// ORACLE -- test infrastructure only.  Nothing in plonky2.5_amd/ may include, link or call this.
//
// CPU restatement of the field arithmetic the plonky2 prover computes in: GoldilocksField
// (p = 2^64 - 2^32 + 1, /root/reference/src/p3/mod.rs:55) and its quadratic extension x^2 = 7
// (/root/reference/src/p3/extension.rs:147-152, 458-471 for the product, 305-321 for the inverse).
// The arithmetic itself lives in the absent third-party crate plonky2_field @ 3de92d9
// (Cargo.toml:15-19); it is restated from the published definition of the field.
// Written deliberately differently from plonky2.5_amd/csrc/gl.h (128-bit integers, always
// canonical) so that the two implementations check each other.
#pragma once
#include <stdint.h>
#include <stddef.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;

static const u64 RP = 0xFFFFFFFF00000001ULL;

static inline u64 rf_reduce(u128 x) {
  // 2^64 = 2^32 - 1, 2^96 = -1 (mod p):  x = x0 + x1*2^64 + x2*2^96 with x1 < 2^32, x2 < 2^32
  // (64-bit carry/borrow form of the identity above -- the shape of upstream's `reduce128` --
  // so that the timed CPU baseline is not handicapped by 128-bit compare/subtract loops)
  u64 x0 = (u64)x;
  u64 x1 = (u64)(x >> 64) & 0xFFFFFFFFULL;
  u64 x2 = (u64)(x >> 96);
  // branch-free: the borrow/carry conditions are data-dependent coin flips, so masks beat jumps
  u64 t0 = x0 - x2;
  t0 -= (0 - (u64)(x0 < x2)) & 0xFFFFFFFFULL;   // borrowed 2^64 = 2^32 - 1 (mod p)
  u64 t1 = (x1 << 32) - x1;                      // x1 * (2^32 - 1)
  u64 r = t0 + t1;
  r += (0 - (u64)(r < t1)) & 0xFFFFFFFFULL;     // carried 2^64
  return r >= RP ? r - RP : r;
}
static inline u64 rf_add(u64 a, u64 b) {
  u128 s = (u128)a + b;
  return (u64)(s >= RP ? s - RP : s);
}
static inline u64 rf_sub(u64 a, u64 b) { return a >= b ? a - b : a + (RP - b); }
static inline u64 rf_neg(u64 a) { return a ? RP - a : 0; }
static inline u64 rf_mul(u64 a, u64 b) { return rf_reduce((u128)a * b); }
static inline u64 rf_pow(u64 b, u64 e) {
  u64 r = 1;
  for (; e; e >>= 1) {
    if (e & 1) r = rf_mul(r, b);
    b = rf_mul(b, b);
  }
  return r;
}
static inline u64 rf_inv(u64 a) { return rf_pow(a, RP - 2); }
static inline u64 rf_root_of_unity(unsigned log_n) {
  u64 g = 1753635133440165772ULL;  // 7^((p-1)/2^32), extension.rs:155 / two_adic.rs:35
  for (unsigned i = log_n; i < 32; i++) g = rf_mul(g, g);
  return g;
}

struct RE2 {
  u64 a, b;
};
static inline RE2 re(u64 a, u64 b = 0) { return RE2{a, b}; }
static inline RE2 re_add(RE2 x, RE2 y) { return RE2{rf_add(x.a, y.a), rf_add(x.b, y.b)}; }
static inline RE2 re_sub(RE2 x, RE2 y) { return RE2{rf_sub(x.a, y.a), rf_sub(x.b, y.b)}; }
static inline RE2 re_neg(RE2 x) { return RE2{rf_neg(x.a), rf_neg(x.b)}; }
static inline RE2 re_mul(RE2 x, RE2 y) {
  return RE2{rf_add(rf_mul(x.a, y.a), rf_mul(7, rf_mul(x.b, y.b))),
             rf_add(rf_mul(x.a, y.b), rf_mul(x.b, y.a))};
}
static inline RE2 re_muls(RE2 x, u64 s) { return RE2{rf_mul(x.a, s), rf_mul(x.b, s)}; }
static inline RE2 re_inv(RE2 x) {
  u64 n = rf_sub(rf_mul(x.a, x.a), rf_mul(7, rf_mul(x.b, x.b)));
  u64 ni = rf_inv(n);
  return RE2{rf_mul(x.a, ni), rf_mul(rf_neg(x.b), ni)};
}
static inline bool re_eq(RE2 x, RE2 y) { return x.a == y.a && x.b == y.b; }
static inline RE2 re_pow(RE2 b, u64 e) {
  RE2 r = re(1);
  for (; e; e >>= 1) {
    if (e & 1) r = re_mul(r, b);
    b = re_mul(b, b);
  }
  return r;
}
static inline RE2 re_exp_pow2(RE2 b, unsigned k) {
  while (k--) b = re_mul(b, b);
  return b;
}
static inline size_t rbits(size_t x, unsigned bits) {
  if (!bits) return 0;
  u64 v = x;
  v = ((v >> 1) & 0x5555555555555555ULL) | ((v & 0x5555555555555555ULL) << 1);
  v = ((v >> 2) & 0x3333333333333333ULL) | ((v & 0x3333333333333333ULL) << 2);
  v = ((v >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((v & 0x0F0F0F0F0F0F0F0FULL) << 4);
  v = __builtin_bswap64(v);
  return (size_t)(v >> (64 - bits));
}
