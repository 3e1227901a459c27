// ORACLE -- test infrastructure only.  The TUNED cpu_baseline leg, second half: the quotient's
// evaluation of the vanishing polynomial on the LDE coset (upstream prover.rs `compute_quotient_polys` ->
// `eval_vanishing_poly_base_batch`, reached from /root/reference/src/p3/mod.rs:260; gate bodies: ref_gates.h, which cites the
// reference's evaluators) for EIGHT coset points at a time: ref_gates.h's evaluators are templates over the field type, and
// FB8 below is the base field on the eight 64-bit lanes of an AVX-512 register -- the same code, the same canonical values,
// eight points per pass (upstream's own `eval_unfiltered_base_batch` / PackedField does the same on its CPU prover).
// Called by ref_quotient_chunks only after p25o_set_tuned(1); the checker runs the scalar FB instantiation.
#include <string.h>
#include <vector>
#include "ref_prover.h"
// This file is compiled with -mavx512f -mavx512dq (oracle/Makefile): ref_gates.h's templates pass field elements by value, and
// a 512-bit value changes the calling convention, so every function that touches FB8 must be compiled for AVX-512 (a
// `#pragma GCC target` region does not cover templates instantiated after it).  The caller checks ref_x8_available() first.
#include "ref_field_x8.h"

struct FB8 {
  V v;
  static FB8 from(u64 x) { return FB8{_mm512_set1_epi64((long long)x)}; }
  FB8 operator+(FB8 o) const { return FB8{v_add(v, o.v)}; }
  FB8 operator-(FB8 o) const { return FB8{v_sub(v, o.v)}; }
  FB8 operator*(FB8 o) const { return FB8{v_mul(v, o.v)}; }
  FB8 smul(u64 s) const { return FB8{v_mul(v, _mm512_set1_epi64((long long)s))}; }
};
#include "ref_gates.h"

// out[k][j] = vanishing polynomial k at coset point i0 + j (natural order), j < 8, BEFORE the division by Z_H
void ref_vanishing_points_x8(const RCircuit& c, const RPolyBatch& constants_sigmas, const RPolyBatch& wires,
                             const RPolyBatch& zs_batch, const u64* betas, const u64* gammas, const u64* alphas,
                             const u64* pih, size_t i0, const u64 x[8], const u64 l0[8], u64 (*out)[8]) {
  const int NC = c.num_challenges, RW = c.num_routed, NP = c.num_partial_products;
  const int lde_bits = c.degree_bits + c.rate_bits;
  const size_t big = (size_t)1 << lde_bits, next_step = (size_t)1 << c.rate_bits;
  const int n_consts = c.num_constants_total();
  long long pos[8], posn[8];
  for (int j = 0; j < 8; j++) {
    pos[j] = (long long)rbits(i0 + j, lde_bits);
    posn[j] = (long long)rbits((i0 + j + next_step) % big, lde_bits);
  }
  auto rows = [&](const RPolyBatch& b, const long long* p) {
    const long long w = (long long)b.n_polys;
    return _mm512_setr_epi64(p[0] * w, p[1] * w, p[2] * w, p[3] * w, p[4] * w, p[5] * w, p[6] * w, p[7] * w);
  };
  const V r_cs = rows(constants_sigmas, pos), r_w = rows(wires, pos), r_z = rows(zs_batch, pos), r_zn = rows(zs_batch, posn);
  auto col = [&](const RPolyBatch& b, V r, int k) {
    return FB8{_mm512_i64gather_epi64(r, (const long long*)(b.leaves_flat.data() + k), 8)};
  };
  std::vector<FB8> consts(n_consts), sig(RW), wv(c.num_wires), z(NC), zn(NC), pp(NC * NP);
  for (int k = 0; k < n_consts; k++) consts[k] = col(constants_sigmas, r_cs, k);
  for (int k = 0; k < RW; k++) sig[k] = col(constants_sigmas, r_cs, n_consts + k);
  for (int k = 0; k < c.num_wires; k++) wv[k] = col(wires, r_w, k);
  for (int k = 0; k < NC; k++) {
    z[k] = col(zs_batch, r_z, k);
    zn[k] = col(zs_batch, r_zn, k);
  }
  for (int k = 0; k < NC * NP; k++) pp[k] = col(zs_batch, r_z, NC + k);
  FB8 outv[8];
  ref_eval_vanishing<FB8>(c, FB8{_mm512_loadu_si512(x)}, FB8{_mm512_loadu_si512(l0)}, consts.data(), sig.data(), wv.data(),
                          z.data(), zn.data(), pp.data(), betas, gammas, alphas, pih, outv);
  for (int k = 0; k < NC; k++) _mm512_storeu_si512(out[k], outv[k].v);
}
