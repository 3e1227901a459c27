// ORACLE -- test infrastructure only.  Nothing in plonky2.5_amd/ may include, link or call this.
//
// Gate constraint evaluators, generic over the field type F (base field for the prover's quotient
// evaluation, F_p^2 for the verifier's check at zeta).  One restatement serves both, as the
// reference's `eval_unfiltered` (extension) and `eval_unfiltered_base_*` agree by construction
// (its `test_eval_fns` unit tests, e.g. poseidon2_gate.rs:575-581, arithmetic_u32.rs:494-499).
//   Poseidon2Gate            /root/reference/src/common/poseidon2/poseidon2_gate.rs:150-231 (233-310)
//   U32ArithmeticGate        src/common/u32/gates/arithmetic_u32.rs:106-165 (303-366)
//   U32InterleaveGate        src/common/u32/gates/interleave_u32.rs:102-142 (250-287)
//   UninterleaveToU32Gate    src/common/u32/gates/uninterleave_to_u32.rs:114-163 (285-335)
//   ArithmeticExtensionGate, PoseidonGate (recursion) and
//   Noop/Constant/PublicInput/BaseSum<2>/Arithmetic/MulExtension/Exponentiation: upstream plonky2 @
//   3de92d9 gates/*.rs (absent crate), restated from SURVEY.md App. A.12 / the published gate definitions.
#pragma once
#include "ref_circuit.h"

struct FB {  // base field element
  u64 v;
  static FB from(u64 x) { return FB{x}; }
  FB operator+(FB o) const { return FB{rf_add(v, o.v)}; }
  FB operator-(FB o) const { return FB{rf_sub(v, o.v)}; }
  FB operator*(FB o) const { return FB{rf_mul(v, o.v)}; }
  FB smul(u64 s) const { return FB{rf_mul(v, s)}; }
  bool is_zero() const { return v == 0; }
};
struct FE {  // quadratic extension element
  RE2 v;
  static FE from(u64 x) { return FE{RE2{x, 0}}; }
  FE operator+(FE o) const { return FE{re_add(v, o.v)}; }
  FE operator-(FE o) const { return FE{re_sub(v, o.v)}; }
  FE operator*(FE o) const { return FE{re_mul(v, o.v)}; }
  FE smul(u64 s) const { return FE{re_muls(v, s)}; }
  bool is_zero() const { return v.a == 0 && v.b == 0; }
};

namespace refp2 {
#include "poseidon2_constants.inc"
}
#define REF_P2_RC refp2::P2_RC
#define REF_P2_RC_MID refp2::P2_RC_MID
#define REF_P2_DIAG_M1 refp2::P2_MAT_DIAG_M_1

namespace refp1 {
static const u64 RC[360] = {
#include "poseidon_constants.inc"
};
static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
}
#define REF_POSEIDON_RC refp1::RC
template <class F>
static void g_poseidon_mds(F s[12]) {  // circ(17,15,41,16,2,28,13,13,39,18,34,20) + diag(8,0,...)
  F o[12];
  for (int r = 0; r < 12; r++) {
    F acc = r == 0 ? s[0].smul(8) : F::from(0);
    for (int i = 0; i < 12; i++) acc = acc + s[(i + r) % 12].smul(refp1::MDS_CIRC[i]);
    o[r] = acc;
  }
  for (int r = 0; r < 12; r++) s[r] = o[r];
}
template <class F>
static F g_pow7(F x) {
  F x2 = x * x, x4 = x2 * x2, x3 = x * x2;
  return x3 * x4;
}
template <class F>
static void g_p2_external(F s[12]) {  // poseidon2.rs:215-243 (field variant)
  for (int b = 0; b < 3; b++) {
    F* x = s + 4 * b;
    F t0 = x[0] + x[1], t1 = x[2] + x[3];
    F t2 = t1 + x[1].smul(2), t3 = t0 + x[3].smul(2);
    F t4 = t3 + t1.smul(4), t5 = t2 + t0.smul(4);
    x[0] = t3 + t5;
    x[1] = t5;
    x[2] = t2 + t4;
    x[3] = t4;
  }
  F st[4];
  for (int l = 0; l < 4; l++) st[l] = s[l] + s[4 + l] + s[8 + l];
  for (int i = 0; i < 12; i++) s[i] = s[i] + st[i % 4];
}
template <class F>
static void g_p2_internal(F s[12]) {  // poseidon2.rs:245-271
  F sum = s[0];
  for (int i = 1; i < 12; i++) sum = sum + s[i];
  for (int i = 0; i < 12; i++) s[i] = s[i].smul(REF_P2_DIAG_M1[i] - 1) + sum;
}

// Evaluates the unfiltered constraints of `kind` on one row.  w: wires, k: the row's 2 constants
// (selector prefix already stripped), pih: public-inputs hash.  Returns the number of constraints.
template <class F>
static int ref_eval_gate(u32 kind, const F* w, const F* k, const F* pih, F* out) {
  int nc = 0;
  const F one = F::from(1);
  switch (kind) {
    case RG_NOOP:
      return 0;
    case RG_CONSTANT:
      out[0] = k[0] - w[0];
      out[1] = k[1] - w[1];
      return 2;
    case RG_PUBLIC_INPUT:
      for (int i = 0; i < 4; i++) out[i] = w[i] - pih[i];
      return 4;
    case RG_BASE_SUM: {
      F acc = F::from(0);
      for (int i = 62; i >= 0; i--) acc = acc.smul(2) + w[1 + i];
      out[nc++] = acc - w[0];
      for (int i = 0; i < 63; i++) out[nc++] = w[1 + i] * (w[1 + i] - one);
      return nc;
    }
    case RG_ARITHMETIC:
      for (int i = 0; i < 20; i++) out[nc++] = w[4 * i + 3] - (w[4 * i] * w[4 * i + 1] * k[0] + w[4 * i + 2] * k[1]);
      return nc;
    case RG_MUL_EXT:
      for (int i = 0; i < 13; i++) {
        const F* a = w + 6 * i;
        const F* b = w + 6 * i + 2;
        const F* o = w + 6 * i + 4;
        F c0 = (a[0] * b[0] + (a[1] * b[1]).smul(7)) * k[0];
        F c1 = (a[0] * b[1] + a[1] * b[0]) * k[0];
        out[nc++] = o[0] - c0;
        out[nc++] = o[1] - c1;
      }
      return nc;
    case RG_EXPONENTIATION: {
      const F base = w[0];
      for (int i = 0; i < 66; i++) {
        F prev = i == 0 ? one : w[68 + i - 1] * w[68 + i - 1];
        F bit = w[1 + (66 - i - 1)];
        F computed = prev * (bit * base + (one - bit));
        out[nc++] = computed - w[68 + i];
      }
      out[nc++] = w[67] - w[68 + 65];
      return nc;
    }
    case RG_U32_ARITHMETIC:
      for (int i = 0; i < 3; i++) {
        F m0 = w[6 * i], m1 = w[6 * i + 1], ad = w[6 * i + 2];
        F lo = w[6 * i + 3], hi = w[6 * i + 4], inv = w[6 * i + 5];
        F computed = m0 * m1 + ad;
        F diff = F::from(0xFFFFFFFFull) - hi;
        F hi_not_max = inv * diff - one;
        out[nc++] = hi_not_max * lo;
        F combined = hi * F::from((u64)1 << 32) + lo;
        out[nc++] = combined - computed;
        F cl = F::from(0), ch = F::from(0);
        for (int j = 31; j >= 0; j--) {
          F limb = w[18 + 32 * i + j];
          out[nc++] = limb * (limb - one) * (limb - F::from(2)) * (limb - F::from(3));
          if (j < 16)
            cl = cl.smul(4) + limb;
          else
            ch = ch.smul(4) + limb;
        }
        out[nc++] = cl - lo;
        out[nc++] = ch - hi;
      }
      return nc;
    case RG_U32_INTERLEAVE:
      for (int i = 0; i < 3; i++) {
        const F* bits = w + 6 + 32 * i;  // big-endian
        F cx = F::from(0), ci = F::from(0);
        for (int b = 0; b < 32; b++) {
          cx = cx.smul(2) + bits[b];
          ci = ci.smul(4) + bits[b];
        }
        out[nc++] = cx - w[2 * i];
        out[nc++] = ci - w[2 * i + 1];
        for (int b = 0; b < 32; b++) out[nc++] = bits[b] * (bits[b] - one);
      }
      return nc;
    case RG_U32_UNINTERLEAVE:
      for (int i = 0; i < 2; i++) {
        const F* bits = w + 6 + 64 * i;
        F cx = F::from(0);
        for (int b = 0; b < 64; b++) cx = cx.smul(2) + bits[b];
        out[nc++] = cx - w[3 * i];
        F ev = F::from(0), od = F::from(0);
        for (int j = 0; j < 32; j++) {
          u64 coeff = (u64)1 << (32 - j - 1);
          ev = ev + bits[2 * j].smul(coeff);
          od = od + bits[2 * j + 1].smul(coeff);
        }
        out[nc++] = ev - w[3 * i + 1];
        out[nc++] = od - w[3 * i + 2];
        for (int b = 0; b < 64; b++) out[nc++] = bits[b] * (bits[b] - one);
      }
      return nc;
    case RG_POSEIDON2: {
      F swap = w[24];
      out[nc++] = swap * (swap - one);
      for (int i = 0; i < 4; i++) out[nc++] = swap * (w[i + 4] - w[i]) - w[25 + i];
      F st[12];
      for (int i = 0; i < 4; i++) {
        st[i] = w[i] + w[25 + i];
        st[i + 4] = w[i + 4] - w[25 + i];
      }
      for (int i = 8; i < 12; i++) st[i] = w[i];
      g_p2_external(st);
      for (int r = 0; r < 4; r++) {
        for (int i = 0; i < 12; i++) st[i] = st[i] + F::from(REF_P2_RC[12 * r + i]);
        if (r != 0)
          for (int i = 0; i < 12; i++) {
            F sb = w[29 + 12 * (r - 1) + i];
            out[nc++] = st[i] - sb;
            st[i] = sb;
          }
        for (int i = 0; i < 12; i++) st[i] = g_pow7(st[i]);
        g_p2_external(st);
      }
      for (int r = 0; r < 22; r++) {
        st[0] = st[0] + F::from(REF_P2_RC_MID[r]);
        F sb = w[65 + r];
        out[nc++] = st[0] - sb;
        st[0] = g_pow7(sb);
        g_p2_internal(st);
      }
      for (int r = 4; r < 8; r++) {
        for (int i = 0; i < 12; i++) st[i] = st[i] + F::from(REF_P2_RC[12 * r + i]);
        for (int i = 0; i < 12; i++) {
          F sb = w[87 + 12 * (r - 4) + i];
          out[nc++] = st[i] - sb;
          st[i] = sb;
        }
        for (int i = 0; i < 12; i++) st[i] = g_pow7(st[i]);
        g_p2_external(st);
      }
      for (int i = 0; i < 12; i++) out[nc++] = st[i] - w[12 + i];
      return nc;
    }
    case RG_ARITH_EXT:  // out - (c0 * m0 * m1 + c1 * addend) over the algebra X^2 = 7, 10 ops of 8 wires
      for (int i = 0; i < 10; i++) {
        const F* a = w + 8 * i;
        const F* b = w + 8 * i + 2;
        const F* ad = w + 8 * i + 4;
        const F* o = w + 8 * i + 6;
        F c0 = (a[0] * b[0] + (a[1] * b[1]).smul(7)) * k[0] + ad[0] * k[1];
        F c1 = (a[0] * b[1] + a[1] * b[0]) * k[0] + ad[1] * k[1];
        out[nc++] = o[0] - c0;
        out[nc++] = o[1] - c1;
      }
      return nc;
    case RG_POSEIDON: {  // upstream gates/poseidon.rs eval_unfiltered, rounds in the naive form (the constraint
                         // polynomials are the same: the fast partial rounds are a linear change of basis)
      F swap = w[24];
      out[nc++] = swap * (swap - one);
      for (int i = 0; i < 4; i++) out[nc++] = swap * (w[i + 4] - w[i]) - w[25 + i];
      F st[12];
      for (int i = 0; i < 4; i++) {
        st[i] = w[i] + w[25 + i];
        st[i + 4] = w[i + 4] - w[25 + i];
      }
      for (int i = 8; i < 12; i++) st[i] = w[i];
      int tr = 29;
      for (int r = 0; r < 30; r++) {
        for (int i = 0; i < 12; i++) st[i] = st[i] + F::from(REF_POSEIDON_RC[12 * r + i]);
        if (r < 4 || r >= 26) {
          if (r != 0)
            for (int i = 0; i < 12; i++) {
              F sb = w[tr++];
              out[nc++] = st[i] - sb;
              st[i] = sb;
            }
          for (int i = 0; i < 12; i++) st[i] = g_pow7(st[i]);
        } else {
          F sb = w[tr++];
          out[nc++] = st[0] - sb;
          st[0] = g_pow7(sb);
        }
        g_poseidon_mds(st);
      }
      for (int i = 0; i < 12; i++) out[nc++] = st[i] - w[12 + i];
      return nc;
    }
    case RG_RANDOM_ACCESS: {  // upstream gates/random_access.rs eval_unfiltered (bits 4, copies 4, 2 extra constants)
      const int bits = 4, vec = 16, copies = 4, routed = (2 + vec) * copies + 2;
      for (int copy = 0; copy < copies; copy++) {
        const F* cw = w + (2 + vec) * copy;   // access_index, claimed_element, list items
        const F* bw = w + routed + bits * copy;
        for (int i = 0; i < bits; i++) out[nc++] = bw[i] * (bw[i] - one);
        F idx = F::from(0);
        for (int i = bits - 1; i >= 0; i--) idx = idx.smul(2) + bw[i];
        out[nc++] = idx - cw[0];
        F items[16];
        for (int i = 0; i < vec; i++) items[i] = cw[2 + i];
        int len = vec;
        for (int bi = 0; bi < bits; bi++) {
          len /= 2;
          for (int i = 0; i < len; i++) items[i] = items[2 * i] + bw[bi] * (items[2 * i + 1] - items[2 * i]);
        }
        out[nc++] = items[0] - cw[1];
      }
      out[nc++] = k[0] - w[(2 + vec) * copies];
      out[nc++] = k[1] - w[(2 + vec) * copies + 1];
      return nc;
    }
    case RG_REDUCING:        // upstream gates/reducing.rs: acc * alpha + coeff - next acc, in the D = 2 algebra
    case RG_REDUCING_EXT: {  // reducing_extension.rs: the same with algebra-valued coefficients
      const bool ext = kind == RG_REDUCING_EXT;
      const int nco = ext ? 32 : 43, cw = ext ? 2 : 1, start_accs = 6 + nco * cw;
      const F al0 = w[2], al1 = w[3];
      F a0 = w[4], a1 = w[5];
      for (int i = 0; i < nco; i++) {
        const int aw = i == nco - 1 ? 0 : start_accs + 2 * i;
        F t0 = a0 * al0 + (a1 * al1).smul(7) + w[6 + cw * i];
        F t1 = a0 * al1 + a1 * al0;
        if (ext) t1 = t1 + w[6 + cw * i + 1];
        out[nc++] = t0 - w[aw];
        out[nc++] = t1 - w[aw + 1];
        a0 = w[aw];
        a1 = w[aw + 1];
      }
      return nc;
    }
    case RG_POSEIDON_MDS:  // upstream gates/poseidon_mds.rs: the MDS layer on 12 algebra elements, component by component
      for (int r = 0; r < 12; r++)
        for (int d = 0; d < 2; d++) {
          F acc = r == 0 ? w[d].smul(8) : F::from(0);
          for (int i = 0; i < 12; i++) acc = acc + w[2 * ((i + r) % 12) + d].smul(refp1::MDS_CIRC[i]);
          out[nc++] = w[24 + 2 * r + d] - acc;
        }
      return nc;
    case RG_COSET_INTERP: {  // upstream gates/coset_interpolation.rs eval_unfiltered, subgroup_bits 4, degree 6
      // wire pairs are elements of the algebra F[X]/(X^2 - 7); the shift and the domain points / weights are scalars
      struct Alg {
        F a, b;
      };
      auto amul = [](Alg x, Alg y) { return Alg{x.a * y.a + (x.b * y.b).smul(7), x.a * y.b + x.b * y.a}; };
      const F shift = w[0];
      const Alg point{w[33], w[34]}, x{w[45], w[46]};
      out[nc++] = point.a - x.a * shift;
      out[nc++] = point.b - x.b * shift;
      const u64 g = rf_root_of_unity(4), inv16 = rf_inv(16);
      Alg eval{F::from(0), F::from(0)}, prod{one, F::from(0)};
      u64 xi = 1;
      for (int c = 0; c < 3; c++) {
        if (c > 0) {
          const Alg ie{w[37 + 2 * (c - 1)], w[38 + 2 * (c - 1)]}, ip{w[41 + 2 * (c - 1)], w[42 + 2 * (c - 1)]};
          out[nc++] = ie.a - eval.a;
          out[nc++] = ie.b - eval.b;
          out[nc++] = ip.a - prod.a;
          out[nc++] = ip.b - prod.b;
          eval = ie;
          prod = ip;
        }
        const int begin = c == 0 ? 0 : 1 + 5 * c, end = c == 0 ? 6 : (1 + 5 * (c + 1) < 16 ? 1 + 5 * (c + 1) : 16);
        for (int i = begin; i < end; i++) {
          const u64 weight = rf_mul(xi, inv16);
          const Alg v{w[1 + 2 * i].smul(weight), w[2 + 2 * i].smul(weight)};
          const Alg term{x.a - F::from(xi), x.b};
          const Alg e1 = amul(eval, term), e2 = amul(v, prod);
          eval = Alg{e1.a + e2.a, e1.b + e2.b};
          prod = amul(prod, term);
          xi = rf_mul(xi, g);
        }
      }
      out[nc++] = w[35] - eval.a;
      out[nc++] = w[36] - eval.b;
      return nc;
    }
    default:
      return 0;
  }
}

// Vanishing polynomial terms reduced with each alpha (upstream vanishing_poly.rs
// eval_vanishing_poly / eval_vanishing_poly_base_batch).
//   consts: num_selectors + num_constants values; sigmas: num_routed; zs/zs_next: num_challenges;
//   pps: num_challenges * num_partial_products; x: evaluation point; l0_x = L_0(x).
template <class F>
static void ref_eval_vanishing(const RCircuit& c, F x, F l0_x, const F* consts, const F* sigmas, const F* wires,
                               const F* zs, const F* zs_next, const F* pps, const u64* betas, const u64* gammas,
                               const u64* alphas, const u64* pih_u64, F* out /*[num_challenges]*/) {
  const int NC = c.num_challenges, RW = c.num_routed, NP = c.num_partial_products, Q = c.quotient_degree_factor;
  std::vector<F> terms;
  terms.reserve(NC * (2 + NP) + c.num_gate_constraints);
  const F one = F::from(1);
  for (int i = 0; i < NC; i++) terms.push_back(l0_x * (zs[i] - one));
  for (int i = 0; i < NC; i++) {
    std::vector<F> num(RW), den(RW);
    for (int j = 0; j < RW; j++) {
      F s_id = x.smul(c.k_is[j]);
      num[j] = wires[j] + s_id.smul(betas[i]) + F::from(gammas[i]);
      den[j] = wires[j] + sigmas[j].smul(betas[i]) + F::from(gammas[i]);
    }
    // check_partial_products: accumulators z_x, pp_0..pp_{NP-1}, z_gx
    for (int ch = 0; ch * Q < RW; ch++) {
      F prev = ch == 0 ? zs[i] : pps[i * NP + ch - 1];
      F next = ch == NP ? zs_next[i] : pps[i * NP + ch];
      F np = one, dp = one;
      for (int j = ch * Q; j < (ch + 1) * Q && j < RW; j++) {
        np = np * num[j];
        dp = dp * den[j];
      }
      terms.push_back(prev * np - next * dp);
    }
  }
  // gate constraints: sum over gate types of filter * constraint
  std::vector<F> gate_terms(c.num_gate_constraints, F::from(0));
  std::vector<F> tmp(c.num_gate_constraints + 8, F::from(0));
  F pih[4];
  for (int i = 0; i < 4; i++) pih[i] = F::from(pih_u64[i]);
  for (size_t gi = 0; gi < c.gates.size(); gi++) {
    const RGateType& g = c.gates[gi];
    F s = consts[g.selector_index];
    F filter = one;
    for (int k = g.group_start; k < g.group_end; k++)
      if (k != (int)gi) filter = filter * (F::from((u64)k) - s);
    if (c.num_selectors > 1) filter = filter * (F::from(0xFFFFFFFFull) - s);
    int ncon = ref_eval_gate<F>(g.kind, wires, consts + c.num_selectors, pih, tmp.data());
    for (int j = 0; j < ncon; j++) gate_terms[j] = gate_terms[j] + filter * tmp[j];
  }
  terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
  for (int i = 0; i < NC; i++) {
    F acc = F::from(0);
    for (size_t t = terms.size(); t-- > 0;) acc = acc.smul(alphas[i]) + terms[t];
    out[i] = acc;
  }
}
