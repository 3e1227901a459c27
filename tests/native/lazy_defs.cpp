// CPU check of the DEFINITIONS in gl_lazy.h (the plain C++ forms the gfx950 sequences are compared with on the device by
// tools/asmcheck.hip): every operation against 128-bit integer arithmetic mod p over all pairs of boundary values and random
// operands, the wrapped decrements inside vanishing products, the five-multiplication extension product, and the lazy
// 16-point network of ntt16.h against the naive DFT.  Prints "LAZY DEFS OK".
#include <stdio.h>
#include <stdlib.h>
#include "ntt16.h"

typedef unsigned __int128 u128;
static const u64 P = gl::P;
static u64 md(u128 x) { return (u64)(x % P); }
static u64 rng_state = 0x9E3779B97F4A7C15ull;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static int bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (bad < 10) { printf(__VA_ARGS__); printf("\n"); } bad++; } } while (0)

static void pair(u64 a, u64 b) {
  const u64 ca = a % P, cb = b % P;
  CHECK(gl::add_nc_c(a, b) % P == md((u128)ca + cb), "add_nc %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
  CHECK(gl::sub_nc_c(a, b) % P == md((u128)ca + P - cb), "sub_nc %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
  if (b <= P) {   // the single-correction forms: second operand at most p (the contract of gl::add_c / gl::sub_c)
    CHECK(gl::add_c_c(a, b) % P == md((u128)ca + cb) && gl::add_c(a, b) == gl::add_c_c(a, b), "add_c %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
    CHECK(gl::sub_c_c(a, b) % P == md((u128)ca + P - cb) && gl::sub_c(a, b) == gl::sub_c_c(a, b), "sub_c %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
  }
  CHECK(gl::mad_nc_s(a, b, a ^ b) % P == md((u128)ca * cb + ((a ^ b) % P)), "mad_nc_s %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
  u64 s, d;
  gl::bfly_nc(a, b, false, s, d);
  CHECK(s % P == md((u128)ca + cb) && d % P == md((u128)ca + P - cb), "bfly %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
  gl::bfly_nc(a, b, true, s, d);
  CHECK(s % P == md((u128)ca + cb) && d % P == md((u128)cb + P - ca), "bfly swap %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
  const gl::E2 x{a, b}, y{b ^ 0x1234567ull, a + 77}, z = gl::e2_mul_nc(x, y), w = gl::mul(gl::E2{ca, cb}, gl::E2{y.a % P, y.b % P});
  CHECK(z.a % P == w.a && z.b % P == w.b, "e2_mul_nc %016llx %016llx", (unsigned long long)a, (unsigned long long)b);
}
static void single(u64 a) {
  const u64 ca = a % P;
  u64 pw = ca;
  for (int e = 0; e < 96; e++) {
    CHECK(gl::shl_nc_c(a, e) % P == pw, "shl_nc %016llx e=%d", (unsigned long long)a, e);
    pw = md((u128)pw * 2);
  }
  // vanishing products on a canonical value: the wrapped factors never change the product
  const u64 l = ca;
  const u64 m1 = md((u128)l + P - 1), m2 = md((u128)l + P - 2), m3 = md((u128)l + P - 3);   // l - k mod p
  CHECK(gl::mul_nc(l, gl::dec_wrap(l, 1)) % P == md((u128)l * m1), "b(b-1) %016llx", (unsigned long long)l);
  const u64 four = gl::mul_nc(gl::mul_nc(l, gl::dec_wrap(l, 1)), gl::mul_nc(gl::dec_wrap(l, 2), gl::dec_wrap(l, 3)));
  const u64 ref = md((u128)md((u128)l * m1) * md((u128)m2 * m3));
  CHECK(four % P == ref, "l(l-1)(l-2)(l-3) %016llx", (unsigned long long)l);
}
template <bool INV>
static void dft(const u64* in) {
  u64 x[16], c[16];
  for (int k = 0; k < 16; k++) { x[k] = in[k]; c[k] = in[k] % P; }
  p25::dft16<INV>(x);
  u64 w = gl::root_of_unity(4);
  if (INV) w = gl::inv(w);
  for (int j = 0; j < 16; j++) {
    u64 acc = 0, wj = gl::pow(w, j), t = 1;
    for (int k = 0; k < 16; k++) { acc = gl::add(acc, gl::mul(c[k], t)); t = gl::mul(t, wj); }
    const int pos = ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3);
    CHECK(x[pos] % P == acc, "dft16<%d> frequency %d", (int)INV, j);
  }
}
int main() {
  const u64 edge[] = {0, 1, 2, 3, 0xFFFFFFFEull, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, 0xFFFFFFFF00000000ull, P - 2, P - 1, P, P + 1, P + 2,
                      ~0ull, ~0ull - 1, ~0ull - 2, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFEFFFFFFFFull, 0xFFFFFFFF00000002ull,
                      0xFFFFFFFFFFFF0000ull, 0xFFFFFFFF0000FFFFull, 0xFFFFFFFE00000000ull};
  const int ne = sizeof(edge) / 8;
  for (int i = 0; i < ne; i++) { single(edge[i]); for (int j = 0; j < ne; j++) pair(edge[i], edge[j]); }
  for (int i = 0; i < 200000; i++) {
    u64 a = rnd(), b = rnd();
    if ((i & 7) == 1) { a |= 0xFFFFFFFF00000000ull; b |= 0xFFFFFFFF00000000ull; }
    if ((i & 7) == 2) { a = ~0ull - (rnd() & 0xFFFFFFFFull); b = ~0ull - (rnd() & 0xFFFFFFFFull); }
    if ((i & 7) == 3) { a &= 0xFFFFFFFFull; b = ~0ull - (rnd() & 0x1FFFFFFFFull); }
    pair(a, b);
    if (i < 4000) single((i & 1) ? a : (a & 7));
  }
  u64 v[16];
  for (int r = 0; r < 3000; r++) {
    for (int k = 0; k < 16; k++) v[k] = (r & 3) == 0 ? edge[(r + 5 * k) % ne] : ((r & 3) == 1 ? (rnd() | 0xFFFFFFFF00000000ull) : rnd());
    dft<false>(v);
    dft<true>(v);
  }
  printf(bad ? "LAZY DEFS FAIL (%d)\n" : "LAZY DEFS OK\n", bad);
  return bad ? 1 : 0;
}
