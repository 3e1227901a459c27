// See recursion.h.
#include "recursion.h"
#include <algorithm>
#include <stdexcept>
#include <string>
#include "poseidon.h"
#include "poseidon2.h"

namespace p25 {

namespace {

// ---------------------------------------------------------------- small helpers (upstream util/reducing.rs semantics)
// sum_i base^i * terms[i]: upstream ReducingFactorTarget::reduce -- up to num_ops + 1 terms as a Horner chain on
// ArithmeticExtensionGate operations (`reduce_arithmetic`), longer lists on ReducingExtensionGate rows of 32
// coefficients each (terms reversed and zero-padded to whole rows, every row's old_acc wired to the previous output).
Ext reduce_ext(CircuitBuilder& b, const std::vector<Ext>& terms, Ext base) {
  if ((int)terms.size() <= gate_info(G_ARITH_EXT).num_ops + 1) {
    Ext acc = b.zero_extension();
    for (size_t i = terms.size(); i-- > 0;) acc = b.mul_add_extension(base, acc, terms[i]);
    return acc;
  }
  std::vector<Ext> rev(terms);
  while (rev.size() % REDX_COEFFS) rev.push_back(b.zero_extension());
  std::reverse(rev.begin(), rev.end());
  Ext acc = b.zero_extension();
  for (size_t off = 0; off < rev.size(); off += REDX_COEFFS) {
    const int row = b.add_gate(G_REDUCING_EXT);
    b.connect_extension(base, Ext{wire(row, 2), wire(row, 3)});
    b.connect_extension(acc, Ext{wire(row, 4), wire(row, 5)});
    for (int i = 0; i < REDX_COEFFS; i++) b.connect_extension(rev[off + i], Ext{wire(row, 6 + 2 * i), wire(row, 7 + 2 * i)});
    acc = Ext{wire(row, 0), wire(row, 1)};
  }
  return acc;
}
// the same for base-field terms (ReducingFactorTarget::reduce_base: ReducingGate rows of 43 coefficients)
Ext reduce_base(CircuitBuilder& b, const std::vector<Target>& terms, Ext base) {
  if ((int)terms.size() <= gate_info(G_ARITH_EXT).num_ops + 1) {
    Ext acc = b.zero_extension();
    for (size_t i = terms.size(); i-- > 0;) acc = b.mul_add_extension(base, acc, b.convert_to_ext(terms[i]));
    return acc;
  }
  std::vector<Target> rev(terms);
  while (rev.size() % RED_COEFFS) rev.push_back(b.zero());
  std::reverse(rev.begin(), rev.end());
  Ext acc = b.zero_extension();
  for (size_t off = 0; off < rev.size(); off += RED_COEFFS) {
    const int row = b.add_gate(G_REDUCING);
    b.connect_extension(base, Ext{wire(row, 2), wire(row, 3)});
    b.connect_extension(acc, Ext{wire(row, 4), wire(row, 5)});
    for (int i = 0; i < RED_COEFFS; i++) b.connect(rev[off + i], wire(row, 6 + i));
    acc = Ext{wire(row, 0), wire(row, 1)};
  }
  return acc;
}
// reduce_with_powers_ext_circuit(terms, alpha: Target)
Ext reduce_with_powers_ext(CircuitBuilder& b, const std::vector<Ext>& terms, Target alpha) {
  return reduce_ext(b, terms, b.convert_to_ext(alpha));
}
Ext cext(CircuitBuilder& b, u64 v) { return b.constant_extension(gl::E2{v, 0}); }
// prod_{i < B} (x - i): the range check of the reference's u32 gates and BaseSumGate
// (acc' = acc * x + (-i) * acc in one arithmetic_extension call: interleave_u32.rs:176-186)
Ext range_product(CircuitBuilder& b, Ext x, int B) {
  Ext acc = b.one_extension();
  for (int i = 0; i < B; i++) acc = b.arithmetic_extension(1, gl::neg((u64)i), acc, x, acc);
  return acc;
}

// ---------------------------------------------------------------- Poseidon2 in-circuit helpers (poseidon2.rs:381-500)
using St = std::array<Ext, 12>;
void matmul_m4_circuit(CircuitBuilder& b, St& s) {  // poseidon2.rs:412-438
  for (int i = 0; i < 3; i++) {
    Ext t0 = b.mul_const_add_extension(1, s[4 * i], s[4 * i + 1]);
    Ext t1 = b.mul_const_add_extension(1, s[4 * i + 2], s[4 * i + 3]);
    Ext t2 = b.mul_const_add_extension(2, s[4 * i + 1], t1);
    Ext t3 = b.mul_const_add_extension(2, s[4 * i + 3], t0);
    Ext t4 = b.mul_const_add_extension(4, t1, t3);
    Ext t5 = b.mul_const_add_extension(4, t0, t2);
    Ext t6 = b.mul_const_add_extension(1, t3, t5);
    Ext t7 = b.mul_const_add_extension(1, t2, t4);
    s[4 * i] = t6;
    s[4 * i + 1] = t5;
    s[4 * i + 2] = t7;
    s[4 * i + 3] = t4;
  }
}
St matmul_external_circuit(CircuitBuilder& b, St& in) {  // poseidon2.rs:382-409
  matmul_m4_circuit(b, in);
  St r;
  for (int blk = 0; blk < 3; blk++)
    for (int l = 0; l < 4; l++) {
      // result[4*blk + l] = in[l] + in[4+l] + in[8+l] + in[4*blk + l], summed in the reference's order
      std::vector<Ext> t;
      if (blk == 0) t = {in[l], in[l], in[4 + l], in[8 + l]};
      if (blk == 1) t = {in[l], in[4 + l], in[4 + l], in[8 + l]};
      if (blk == 2) t = {in[l], in[4 + l], in[8 + l], in[8 + l]};
      r[4 * blk + l] = b.add_many_extension(t);
    }
  return r;
}
void constant_layer_circuit(CircuitBuilder& b, St& s, int r) {  // poseidon2.rs:441-453
  for (int i = 0; i < 12; i++) s[i] = b.add_extension(s[i], cext(b, poseidon2::P2_RC[12 * r + i]));
}
void sbox_layer_circuit(CircuitBuilder& b, St& s) {  // poseidon2.rs:456-465
  for (int i = 0; i < 12; i++) s[i] = b.exp_u64_extension(s[i], 7);
}
void matmul_internal_circuit(CircuitBuilder& b, St& s) {  // poseidon2.rs:479-499
  Ext sum = b.add_many_extension(std::vector<Ext>(s.begin(), s.end()));
  for (int i = 0; i < 12; i++) s[i] = b.mul_add_extension(cext(b, poseidon2::P2_MAT_DIAG_M_1[i] - 1), s[i], sum);
}

}  // namespace

// ---------------------------------------------------------------- gate evaluators in-circuit
std::vector<Ext> eval_gate_circuit(CircuitBuilder& b, GateKind kind, const std::vector<Ext>& w, const Ext k[2],
                                   const std::array<Target, 4>& pih) {
  std::vector<Ext> c;
  const Ext one = b.one_extension();
  switch (kind) {
    case G_NOOP:
      break;
    case G_CONSTANT:  // const_i - wire_i
      for (int i = 0; i < 2; i++) c.push_back(b.sub_extension(k[i], w[i]));
      break;
    case G_PUBLIC_INPUT:  // wire_i - public_inputs_hash[i]
      for (int i = 0; i < 4; i++) c.push_back(b.sub_extension(w[i], b.convert_to_ext(pih[i])));
      break;
    case G_BASE_SUM: {  // sum_i limb_i 2^i - sum, then limb (limb - 1) per limb
      std::vector<Ext> limbs(w.begin() + 1, w.begin() + 1 + BASE_SUM_LIMBS);
      Ext computed = reduce_with_powers_ext(b, limbs, b.constant(2));
      c.push_back(b.sub_extension(computed, w[0]));
      for (const Ext& l : limbs) c.push_back(range_product(b, l, 2));
      break;
    }
    case G_ARITHMETIC:  // output - (c0 * m0 * m1 + c1 * addend)
      for (int i = 0; i < 20; i++) {
        Ext scaled = b.mul_many_extension({k[0], w[4 * i], w[4 * i + 1]});
        Ext computed = b.mul_add_extension(k[1], w[4 * i + 2], scaled);
        c.push_back(b.sub_extension(w[4 * i + 3], computed));
      }
      break;
    case G_MUL_EXT:      // the row's wire pairs are elements of the algebra F_ext[X]/(X^2 - 7)
    case G_ARITH_EXT: {
      const int ops = kind == G_MUL_EXT ? 13 : 10, stride = kind == G_MUL_EXT ? 6 : 8;
      for (int i = 0; i < ops; i++) {
        const Ext* a = &w[stride * i];
        const Ext* m = &w[stride * i + 2];
        // (a0 + a1 X)(m0 + m1 X) = a0 m0 + 7 a1 m1 + (a0 m1 + a1 m0) X
        Ext p0 = b.mul_add_extension(a[0], m[0], b.mul_const_extension(7, b.mul_extension(a[1], m[1])));
        Ext p1 = b.mul_add_extension(a[0], m[1], b.mul_extension(a[1], m[0]));
        Ext r0 = b.mul_extension(k[0], p0), r1 = b.mul_extension(k[0], p1);
        const Ext* out = &w[stride * i + (kind == G_MUL_EXT ? 4 : 6)];
        if (kind == G_ARITH_EXT) {
          r0 = b.mul_add_extension(k[1], w[stride * i + 4], r0);
          r1 = b.mul_add_extension(k[1], w[stride * i + 5], r1);
        }
        c.push_back(b.sub_extension(out[0], r0));
        c.push_back(b.sub_extension(out[1], r1));
      }
      break;
    }
    case G_EXPONENTIATION: {  // prev * (bit * base + 1 - bit) - intermediate_i; output - last intermediate
      const Ext base = w[0];
      for (int i = 0; i < EXP_POWER_BITS; i++) {
        Ext prev = i == 0 ? one : b.square_extension(w[2 + EXP_POWER_BITS + i - 1]);
        Ext bit = w[1 + (EXP_POWER_BITS - 1 - i)];
        Ext not_bit = b.sub_extension(one, bit);
        Ext mul_by = b.mul_add_extension(bit, base, not_bit);
        c.push_back(b.mul_sub_extension(prev, mul_by, w[2 + EXP_POWER_BITS + i]));
      }
      c.push_back(b.sub_extension(w[1 + EXP_POWER_BITS], w[2 + EXP_POWER_BITS + EXP_POWER_BITS - 1]));
      break;
    }
    case G_U32_ARITHMETIC:  // arithmetic_u32.rs:178-245
      for (int i = 0; i < 3; i++) {
        Ext computed = b.mul_add_extension(w[6 * i], w[6 * i + 1], w[6 * i + 2]);
        Ext lo = w[6 * i + 3], hi = w[6 * i + 4], inv = w[6 * i + 5];
        Ext diff = b.sub_extension(cext(b, 0xFFFFFFFFull), hi);
        Ext hi_not_max = b.mul_sub_extension(inv, diff, one);
        c.push_back(b.mul_extension(hi_not_max, lo));
        Ext combined = b.mul_add_extension(hi, cext(b, (u64)1 << 32), lo);
        c.push_back(b.sub_extension(combined, computed));
        Ext cl = b.zero_extension(), ch = b.zero_extension();
        const Ext four = cext(b, 4);
        for (int j = 31; j >= 0; j--) {
          Ext limb = w[18 + 32 * i + j];
          Ext product = one;
          for (int x = 0; x < 4; x++) product = b.mul_extension(product, b.sub_extension(limb, cext(b, (u64)x)));
          c.push_back(product);
          if (j < 16)
            cl = b.mul_add_extension(four, cl, limb);
          else
            ch = b.mul_add_extension(four, ch, limb);
        }
        c.push_back(b.sub_extension(cl, lo));
        c.push_back(b.sub_extension(ch, hi));
      }
      break;
    case G_U32_INTERLEAVE:  // interleave_u32.rs:143-189
      for (int i = 0; i < 3; i++) {
        std::vector<Ext> bits(w.begin() + 6 + 32 * i, w.begin() + 6 + 32 * (i + 1));  // big-endian
        std::vector<Ext> rev(bits.rbegin(), bits.rend());
        c.push_back(b.sub_extension(reduce_with_powers_ext(b, rev, b.constant(2)), w[2 * i]));
        c.push_back(b.sub_extension(reduce_with_powers_ext(b, rev, b.constant(4)), w[2 * i + 1]));
        for (const Ext& bit : bits) c.push_back(range_product(b, bit, 2));
      }
      break;
    case G_U32_UNINTERLEAVE:  // uninterleave_to_u32.rs:164-228
      for (int i = 0; i < 2; i++) {
        std::vector<Ext> bits(w.begin() + 6 + 64 * i, w.begin() + 6 + 64 * (i + 1));
        std::vector<Ext> rev(bits.rbegin(), bits.rend());
        c.push_back(b.sub_extension(reduce_with_powers_ext(b, rev, b.constant(2)), w[3 * i]));
        Ext ev = b.zero_extension(), od = b.zero_extension();
        for (int j = 0; j < 32; j++) {
          Ext coeff = cext(b, (u64)1 << (32 - j - 1));
          ev = b.mul_add_extension(coeff, bits[2 * j], ev);
          od = b.mul_add_extension(coeff, bits[2 * j + 1], od);
        }
        c.push_back(b.sub_extension(ev, w[3 * i + 1]));
        c.push_back(b.sub_extension(od, w[3 * i + 2]));
        for (const Ext& bit : bits) c.push_back(range_product(b, bit, 2));
      }
      break;
    case G_POSEIDON2: {  // poseidon2_gate.rs:312-397
      Ext swap = w[24];
      c.push_back(b.mul_sub_extension(swap, swap, swap));
      for (int i = 0; i < 4; i++) {
        Ext diff = b.sub_extension(w[i + 4], w[i]);
        c.push_back(b.mul_sub_extension(swap, diff, w[25 + i]));
      }
      St st;
      for (int i = 0; i < 4; i++) {
        st[i] = b.add_extension(w[i], w[25 + i]);
        st[i + 4] = b.sub_extension(w[i + 4], w[25 + i]);
      }
      for (int i = 8; i < 12; i++) st[i] = w[i];
      st = matmul_external_circuit(b, st);
      for (int r = 0; r < 4; r++) {
        constant_layer_circuit(b, st, r);
        if (r != 0)
          for (int i = 0; i < 12; i++) {
            Ext sb = w[29 + 12 * (r - 1) + i];
            c.push_back(b.sub_extension(st[i], sb));
            st[i] = sb;
          }
        sbox_layer_circuit(b, st);
        st = matmul_external_circuit(b, st);
      }
      for (int r = 0; r < 22; r++) {
        st[0] = b.add_extension(st[0], cext(b, poseidon2::P2_RC_MID[r]));
        Ext sb = w[65 + r];
        c.push_back(b.sub_extension(st[0], sb));
        st[0] = b.exp_u64_extension(sb, 7);
        matmul_internal_circuit(b, st);
      }
      for (int r = 4; r < 8; r++) {
        constant_layer_circuit(b, st, r);
        for (int i = 0; i < 12; i++) {
          Ext sb = w[87 + 12 * (r - 4) + i];
          c.push_back(b.sub_extension(st[i], sb));
          st[i] = sb;
        }
        sbox_layer_circuit(b, st);
        st = matmul_external_circuit(b, st);
      }
      for (int i = 0; i < 12; i++) c.push_back(b.sub_extension(st[i], w[12 + i]));
      break;
    }
    case G_RANDOM_ACCESS: {  // upstream gates/random_access.rs eval_unfiltered_circuit
      const Ext zero = b.zero_extension(), two = cext(b, 2);
      for (int copy = 0; copy < RA_COPIES; copy++) {
        const int base = (2 + RA_VEC) * copy;
        std::vector<Ext> bits, items;
        for (int i = 0; i < RA_BITS; i++) bits.push_back(w[RA_ROUTED + RA_BITS * copy + i]);
        for (int i = 0; i < RA_VEC; i++) items.push_back(w[base + 2 + i]);
        for (const Ext& bt : bits) c.push_back(b.mul_sub_extension(bt, bt, bt));
        Ext idx = zero;
        for (int i = RA_BITS - 1; i >= 0; i--) idx = b.mul_add_extension(idx, two, bits[i]);
        c.push_back(b.sub_extension(idx, w[base]));
        for (const Ext& bt : bits) {   // select_ext_generalized(bit, y, x) on adjacent pairs
          std::vector<Ext> nxt;
          for (size_t i = 0; i + 1 < items.size(); i += 2) {
            Ext tmp = b.mul_sub_extension(bt, items[i], items[i]);
            nxt.push_back(b.mul_sub_extension(bt, items[i + 1], tmp));
          }
          items.swap(nxt);
        }
        c.push_back(b.sub_extension(items[0], w[base + 1]));
      }
      for (int i = 0; i < RA_EXTRA_CONSTS; i++) c.push_back(b.sub_extension(k[i], w[(2 + RA_VEC) * RA_COPIES + i]));
      break;
    }
    case G_REDUCING:
    case G_REDUCING_EXT: {  // upstream gates/reducing{,_extension}.rs eval_unfiltered_circuit: acc * alpha + coeff - next acc
      const bool ext = kind == G_REDUCING_EXT;                       // in the algebra F_ext[X]/(X^2 - 7) over wire pairs
      const int nco = ext ? REDX_COEFFS : RED_COEFFS, cw = ext ? 2 : 1, start_accs = 6 + nco * cw;
      const Ext al0 = w[2], al1 = w[3];
      Ext a0 = w[4], a1 = w[5];
      for (int i = 0; i < nco; i++) {
        const int aw = i == nco - 1 ? 0 : start_accs + 2 * i;
        // (a0 + a1 X)(al0 + al1 X) + coeff
        Ext t0 = b.mul_add_extension(a0, al0, b.mul_const_add_extension(7, b.mul_extension(a1, al1), w[6 + cw * i]));
        Ext t1 = b.mul_add_extension(a0, al1, b.mul_extension(a1, al0));
        if (ext) t1 = b.add_extension(t1, w[6 + cw * i + 1]);
        c.push_back(b.sub_extension(t0, w[aw]));
        c.push_back(b.sub_extension(t1, w[aw + 1]));
        a0 = w[aw];
        a1 = w[aw + 1];
      }
      break;
    }
    case G_COSET_INTERP: {  // upstream gates/coset_interpolation.rs eval_unfiltered_circuit: the recurrence in the algebra
      using Alg = std::array<Ext, 2>;                       // F_ext[X]/(X^2 - 7) over wire pairs
      auto amul = [&](const Alg& x, const Alg& y) {
        return Alg{b.mul_add_extension(x[0], y[0], b.mul_const_extension(7, b.mul_extension(x[1], y[1]))),
                   b.mul_add_extension(x[0], y[1], b.mul_extension(x[1], y[0]))};
      };
      const Ext shift = w[0];
      const Alg point{w[CI_W_POINT], w[CI_W_POINT + 1]}, x{w[CI_W_SHIFTED], w[CI_W_SHIFTED + 1]};
      for (int d = 0; d < 2; d++) c.push_back(b.sub_extension(point[d], b.mul_extension(x[d], shift)));
      const u64 g = gl::root_of_unity(4), inv16 = gl::inv(16);
      Alg eval{b.zero_extension(), b.zero_extension()}, prod{one, b.zero_extension()};
      u64 xi = 1;
      for (int ch = 0; ch <= CI_INTER; ch++) {
        if (ch > 0) {
          const Alg ie{w[CI_W_INTER + 2 * (ch - 1)], w[CI_W_INTER + 2 * (ch - 1) + 1]};
          const Alg ip{w[CI_W_INTER + 2 * (CI_INTER + ch - 1)], w[CI_W_INTER + 2 * (CI_INTER + ch - 1) + 1]};
          for (int d = 0; d < 2; d++) c.push_back(b.sub_extension(ie[d], eval[d]));
          for (int d = 0; d < 2; d++) c.push_back(b.sub_extension(ip[d], prod[d]));
          eval = ie;
          prod = ip;
        }
        for (int i = ci_chunk_begin(ch); i < ci_chunk_end(ch); i++) {
          const u64 weight = gl::mul(xi, inv16);
          const Alg v{b.mul_const_extension(weight, w[1 + 2 * i]), b.mul_const_extension(weight, w[2 + 2 * i])};
          const Alg term{b.sub_extension(x[0], cext(b, xi)), x[1]};
          const Alg e1 = amul(eval, term), e2 = amul(v, prod);
          eval = Alg{b.add_extension(e1[0], e2[0]), b.add_extension(e1[1], e2[1])};
          prod = amul(prod, term);
          xi = gl::mul(xi, g);
        }
      }
      for (int d = 0; d < 2; d++) c.push_back(b.sub_extension(w[CI_W_VALUE + d], eval[d]));
      break;
    }
    case G_POSEIDON_MDS:  // upstream gates/poseidon_mds.rs eval_unfiltered_circuit (mds_layer_algebra_circuit)
      for (int r = 0; r < 12; r++)
        for (int d = 0; d < 2; d++) {
          Ext acc = r == 0 ? b.mul_const_extension(poseidon::MDS_DIAG0, w[d]) : b.zero_extension();
          for (int i = 0; i < 12; i++) acc = b.mul_const_add_extension(poseidon::MDS_CIRC[i], w[2 * ((i + r) % 12) + d], acc);
          c.push_back(b.sub_extension(w[24 + 2 * r + d], acc));
        }
      break;
    case G_POSEIDON: {  // upstream gates/poseidon.rs eval_unfiltered_circuit, rounds in the defining form (upstream: fast
                        // partial rounds), the MDS layer of every round on a PoseidonMdsGate row as upstream's
      Ext swap = w[24];
      c.push_back(b.mul_sub_extension(swap, swap, swap));
      for (int i = 0; i < 4; i++) {
        Ext diff = b.sub_extension(w[i + 4], w[i]);
        c.push_back(b.mul_sub_extension(swap, diff, w[25 + i]));
      }
      St st;
      for (int i = 0; i < 4; i++) {
        st[i] = b.add_extension(w[i], w[25 + i]);
        st[i + 4] = b.sub_extension(w[i + 4], w[25 + i]);
      }
      for (int i = 8; i < 12; i++) st[i] = w[i];
      int tr = 29;
      for (int r = 0; r < poseidon::N_ROUNDS; r++) {
        for (int i = 0; i < 12; i++) st[i] = b.add_extension(st[i], cext(b, poseidon::RC[12 * r + i]));
        if (r < poseidon::HALF_FULL || r >= poseidon::HALF_FULL + poseidon::N_PARTIAL) {
          if (r != 0)
            for (int i = 0; i < 12; i++) {
              Ext sb = w[tr++];
              c.push_back(b.sub_extension(st[i], sb));
              st[i] = sb;
            }
          for (int i = 0; i < 12; i++) st[i] = b.exp_u64_extension(st[i], 7);
        } else {
          Ext sb = w[tr++];
          c.push_back(b.sub_extension(st[0], sb));
          st[0] = b.exp_u64_extension(sb, 7);
        }
        st = b.poseidon_mds_layer(st);   // upstream `mds_layer_circuit`: one PoseidonMdsGate row (48 wires <= 80 routed)
      }
      for (int i = 0; i < 12; i++) c.push_back(b.sub_extension(st[i], w[12 + i]));
      break;
    }
    default:
      throw std::invalid_argument("eval_gate_circuit: gate has no in-circuit evaluator (only gates of the inner "
                                  "plonky3-verifier circuits are supported)");
  }
  if ((int)c.size() != gate_info(kind).num_constraints) throw std::logic_error("eval_gate_circuit: constraint count");
  return c;
}

Circuit build_gate_eval_circuit(GateKind kind) {
  if (kind >= G_NUM_KINDS) throw std::invalid_argument("no in-circuit evaluator for this gate");
  CircuitBuilder cb;
  auto in = [&]() {
    Target t = cb.add_virtual_target();
    cb.input_targets.push_back(t);
    return t;
  };
  std::vector<Ext> wires(cb.config.num_wires);
  for (auto& e : wires) e = Ext{in(), in()};
  Ext consts[2];
  for (auto& e : consts) e = Ext{in(), in()};
  std::array<Target, 4> pih;
  for (auto& t : pih) t = in();
  std::vector<Ext> cons = eval_gate_circuit(cb, kind, wires, consts, pih);
  for (const Ext& e : cons) cb.connect_extension(e, Ext{in(), in()});
  return cb.build();
}

// ======================================================================================================================
// The recursive verifier
// ======================================================================================================================
namespace {

using Hash = std::array<Target, 4>;

// upstream iop/challenger.rs RecursiveChallenger: inputs are buffered and absorbed (overwrite mode, rate 8) when the
// next challenge is drawn; challenges are popped from the end of the 8-word output.
struct RecursiveChallenger {
  CircuitBuilder& b;
  std::array<Target, 12> state;
  std::vector<Target> in, out;
  explicit RecursiveChallenger(CircuitBuilder& cb) : b(cb) {
    for (auto& t : state) t = b.zero();
  }
  void observe(Target t) {
    out.clear();
    in.push_back(t);
  }
  void observe_hash(const Hash& h) {
    for (Target t : h) observe(t);
  }
  void observe_cap(const std::vector<Hash>& cap) {
    for (const Hash& h : cap) observe_hash(h);
  }
  void observe_ext(Ext e) {
    observe(e[0]);
    observe(e[1]);
  }
  void absorb() {
    if (in.empty()) return;
    for (size_t off = 0; off < in.size(); off += 8) {
      for (size_t i = 0; i < 8 && off + i < in.size(); i++) state[i] = in[off + i];
      state = b.poseidon_permute(state);
    }
    out.assign(state.begin(), state.begin() + 8);
    in.clear();
  }
  Target challenge() {
    absorb();
    if (out.empty()) {
      state = b.poseidon_permute(state);
      out.assign(state.begin(), state.begin() + 8);
    }
    Target t = out.back();
    out.pop_back();
    return t;
  }
  Ext ext_challenge() {
    Target a = challenge();
    Target c = challenge();
    return Ext{a, c};
  }
};

// The inner proof as targets, read off the flat layout (include/p25.h "Proof layout").
struct ProofTargets {
  std::vector<Hash> wires_cap, zs_cap, quotient_cap;
  std::vector<Ext> constants, sigmas, wires, zs, zs_next, pps, quotient;
  std::vector<std::vector<Hash>> fri_caps;
  struct Query {
    std::vector<Target> leaf[4];
    std::vector<Hash> path[4];
    std::vector<std::vector<Ext>> step_evals;
    std::vector<std::vector<Hash>> step_path;
  };
  std::vector<Query> queries;
  std::vector<Ext> final_poly;
  Target pow_witness;
  std::vector<Target> public_inputs;
};

struct Reader {
  CircuitBuilder& b;
  size_t count = 0;
  Target word() {
    Target t = b.add_virtual_target();
    b.input_targets.push_back(t);
    count++;
    return t;
  }
  Hash hash() { return Hash{word(), word(), word(), word()}; }
  Ext ext() {
    Target a = word();
    Target c = word();
    return Ext{a, c};
  }
  std::vector<Hash> hashes(size_t n) {
    std::vector<Hash> v(n);
    for (auto& h : v) h = hash();
    return v;
  }
  std::vector<Ext> exts(size_t n) {
    std::vector<Ext> v(n);
    for (auto& e : v) e = ext();
    return v;
  }
};

ProofTargets add_virtual_proof(CircuitBuilder& b, const Circuit& c) {
  Reader r{b};
  ProofTargets p;
  const size_t capn = (size_t)1 << c.cfg.cap_height;
  const int NC = c.cfg.num_challenges, NP = c.num_partial_products;
  const int width[4] = {(int)c.constants_sigmas.size(), c.cfg.num_wires, NC * (1 + NP), NC * c.cfg.max_quotient_degree_factor};
  p.wires_cap = r.hashes(capn);
  p.zs_cap = r.hashes(capn);
  p.quotient_cap = r.hashes(capn);
  p.constants = r.exts(c.num_selectors + c.cfg.num_constants);
  p.sigmas = r.exts(c.cfg.num_routed_wires);
  p.wires = r.exts(c.cfg.num_wires);
  p.zs = r.exts(NC);
  p.zs_next = r.exts(NC);
  p.pps = r.exts((size_t)NC * NP);
  p.quotient = r.exts(width[3]);
  for (size_t l = 0; l < c.fri_reduction_arity_bits.size(); l++) p.fri_caps.push_back(r.hashes(capn));
  const int lde_bits = c.degree_bits + c.cfg.rate_bits;
  p.queries.resize(c.cfg.num_query_rounds);
  for (auto& q : p.queries) {
    for (int o = 0; o < 4; o++) {
      for (int k = 0; k < width[o]; k++) q.leaf[o].push_back(r.word());
      q.path[o] = r.hashes(lde_bits - c.cfg.cap_height);
    }
    int bits = lde_bits;
    for (int a : c.fri_reduction_arity_bits) {
      bits -= a;
      q.step_evals.push_back(r.exts((size_t)1 << a));
      q.step_path.push_back(r.hashes(bits - c.cfg.cap_height));
    }
  }
  int fdeg = c.degree_bits;
  for (int a : c.fri_reduction_arity_bits) fdeg -= a;
  p.final_poly = r.exts((size_t)1 << fdeg);
  p.pow_witness = r.word();
  for (size_t i = 0; i < c.public_inputs.size(); i++) p.public_inputs.push_back(r.word());  // ProofWithPublicInputs::public_inputs
  return p;
}

// upstream hash/merkle_proofs.rs verify_merkle_proof_to_cap_with_cap_index; the cap entry is selected with a
// one-hot vector of the cap index (instead of upstream's RandomAccessGate).
void verify_merkle_proof_to_cap(CircuitBuilder& b, const std::vector<Target>& leaf, const std::vector<BoolTarget>& index_bits,
                                Target cap_index, const std::vector<Target>& cap_onehot, const std::vector<Hash>& cap,
                                const std::vector<Hash>& siblings) {
  Hash state = b.hash_or_noop(leaf);
  if (siblings.size() > index_bits.size()) throw std::logic_error("merkle: more siblings than index bits");
  for (size_t k = 0; k < siblings.size(); k++) {
    std::array<Target, 12> in;
    for (int i = 0; i < 4; i++) {
      in[i] = state[i];
      in[4 + i] = siblings[k][i];
      in[8 + i] = b.zero();
    }
    auto out = b.poseidon_permute_swapped(in, index_bits[k]);
    state = Hash{out[0], out[1], out[2], out[3]};
  }
  if ((int)cap.size() == RA_VEC || cap.size() == 1) {
    // upstream: `let state_cap = self.random_access_hash(cap_index, merkle_cap.0.clone()); connect_hashes(state, ..)`
    Hash sel = b.random_access_hash(cap_index, cap);
    for (int i = 0; i < 4; i++) b.connect(sel[i], state[i]);
    return;
  }
  for (int i = 0; i < 4; i++) {   // caps of other sizes (no RandomAccessGate instance for them here): one-hot select
    Target sel = b.zero();
    for (size_t k = 0; k < cap.size(); k++) sel = b.mul_add(cap_onehot[k], cap[k][i], sel);
    b.connect(sel, state[i]);
  }
}
// one-hot of the integer with little-endian `bits`: out[k] = prod_j (bit_j if k_j else 1 - bit_j)
std::vector<Target> one_hot(CircuitBuilder& b, const std::vector<BoolTarget>& bits) {
  std::vector<Target> v = {b.one()};
  for (size_t j = bits.size(); j-- > 0;) {  // most significant first so that index = sum bit_j 2^j
    std::vector<Target> nxt(v.size() * 2);
    for (size_t k = 0; k < v.size(); k++) {
      Target hi = b.mul(v[k], bits[j]);
      nxt[2 * k] = b.sub(v[k], hi);
      nxt[2 * k + 1] = hi;
    }
    v.swap(nxt);
  }
  return v;
}
// upstream exp_from_bits_const_base: prod_i (bit_i ? base^(2^i) : 1), bits little-endian
Target exp_from_bits_const_base(CircuitBuilder& b, u64 base, const std::vector<BoolTarget>& bits) {
  Target product = b.one();
  u64 pw = base;
  for (size_t i = 0; i < bits.size(); i++) {
    // product *= 1 + bit (base^(2^i) - 1)
    product = b.arithmetic(gl::sub(pw, 1), 1, product, bits[i], product);
    pw = gl::mul(pw, pw);
  }
  return product;
}

// one inner proof; returns its targets (the caller may expose some of them as public inputs of the outer circuit)
ProofTargets verify_one(CircuitBuilder& b, const Circuit& c, const Hash& digest, const std::vector<Hash>& cs_cap) {
  ProofTargets p = add_virtual_proof(b, c);
  const int NC = c.cfg.num_challenges, NP = c.num_partial_products, RW = c.cfg.num_routed_wires;
  const int Q = c.cfg.max_quotient_degree_factor;
  const int db = c.degree_bits, rb = c.cfg.rate_bits, lde_bits = db + rb;
  // "let public_inputs_hash = self.hash_n_to_hash_no_pad::<C::InnerHasher>(proof_with_pis.public_inputs)"
  // (plonk/recursive_verifier.rs verify_proof); of the empty list it is four zeros and costs no row
  const Hash pih = b.hash_n_to_hash_no_pad(p.public_inputs);

  // ---- challenges (upstream plonk/get_challenges.rs, in-circuit)
  RecursiveChallenger ch(b);
  ch.observe_hash(digest);
  ch.observe_hash(pih);
  ch.observe_cap(p.wires_cap);
  std::vector<Target> betas, gammas, alphas;
  for (int i = 0; i < NC; i++) betas.push_back(ch.challenge());
  for (int i = 0; i < NC; i++) gammas.push_back(ch.challenge());
  ch.observe_cap(p.zs_cap);
  for (int i = 0; i < NC; i++) alphas.push_back(ch.challenge());
  ch.observe_cap(p.quotient_cap);
  const Ext zeta = ch.ext_challenge();
  for (auto* v : {&p.constants, &p.sigmas, &p.wires, &p.zs, &p.pps, &p.quotient, &p.zs_next})
    for (const Ext& e : *v) ch.observe_ext(e);
  const Ext fri_alpha = ch.ext_challenge();
  std::vector<Ext> fri_betas;
  for (auto& cap : p.fri_caps) {
    ch.observe_cap(cap);
    fri_betas.push_back(ch.ext_challenge());
  }
  for (const Ext& e : p.final_poly) ch.observe_ext(e);
  ch.observe(p.pow_witness);
  const Target pow_response = ch.challenge();
  std::vector<Target> query_indices;
  for (int i = 0; i < c.cfg.num_query_rounds; i++) query_indices.push_back(ch.challenge());

  // ---- vanishing(zeta) == Z_H(zeta) * sum_k zeta^(n k) t_k(zeta)   (plonk/recursive_verifier.rs, vanishing_poly.rs)
  const Ext one = b.one_extension();
  const Ext zeta_pow_deg = b.exp_power_of_2_extension(zeta, db);
  {
    // L_0(zeta) = (zeta^n - 1) / (n (zeta - 1))
    Ext zero_poly = b.sub_extension(zeta_pow_deg, one);
    const u64 nf = (u64)1 << db;
    Ext denominator = b.arithmetic_extension(nf, nf, zeta, one, b.convert_to_ext(b.neg_one()));
    Ext l0 = b.div_extension(zero_poly, denominator);
    std::vector<Ext> z1_terms, pp_terms;
    std::vector<Ext> s_ids;
    for (int j = 0; j < RW; j++) s_ids.push_back(b.scalar_mul_ext(b.constant(c.k_is[j]), zeta));
    for (int i = 0; i < NC; i++) {
      z1_terms.push_back(b.mul_sub_extension(l0, p.zs[i], l0));
      std::vector<Ext> num, den;
      Ext beta = b.convert_to_ext(betas[i]), gamma = b.convert_to_ext(gammas[i]);
      for (int j = 0; j < RW; j++) {
        Ext wg = b.add_extension(p.wires[j], gamma);
        num.push_back(b.mul_add_extension(beta, s_ids[j], wg));
        den.push_back(b.mul_add_extension(beta, p.sigmas[j], wg));
      }
      // check_partial_products_circuit: accumulators Z(x), pp_0 .. pp_{NP-1}, Z(g x); chunks of Q routed wires
      for (int k = 0; k * Q < RW; k++) {
        Ext prev = k == 0 ? p.zs[i] : p.pps[(size_t)i * NP + k - 1];
        Ext next = k == NP ? p.zs_next[i] : p.pps[(size_t)i * NP + k];
        std::vector<Ext> nc(num.begin() + k * Q, num.begin() + std::min(RW, (k + 1) * Q));
        std::vector<Ext> dc(den.begin() + k * Q, den.begin() + std::min(RW, (k + 1) * Q));
        Ext np = b.mul_many_extension(nc), dp = b.mul_many_extension(dc);
        Ext next_dp = b.mul_extension(next, dp);
        pp_terms.push_back(b.mul_sub_extension(prev, np, next_dp));
      }
    }
    // evaluate_gate_constraints_circuit: sum over gate types of filter * constraint
    std::vector<Ext> gate_terms(c.num_gate_constraints, b.zero_extension());
    const Ext consts[2] = {p.constants[c.num_selectors], p.constants[c.num_selectors + 1]};
    for (size_t gi = 0; gi < c.gates.size(); gi++) {
      const int si = c.selector_index[gi];
      const Ext s = p.constants[si];
      std::vector<Ext> factors;
      for (int k = c.groups[si].first; k < c.groups[si].second; k++)
        if (k != (int)gi) factors.push_back(b.sub_extension(cext(b, (u64)k), s));
      if (c.num_selectors > 1) factors.push_back(b.sub_extension(cext(b, 0xFFFFFFFFull), s));  // UNUSED_SELECTOR
      Ext filter = b.mul_many_extension(factors);
      std::vector<Ext> cons = eval_gate_circuit(b, c.gates[gi], p.wires, consts, pih);
      for (size_t j = 0; j < cons.size(); j++) gate_terms[j] = b.mul_add_extension(filter, cons[j], gate_terms[j]);
    }
    std::vector<Ext> terms(z1_terms);
    terms.insert(terms.end(), pp_terms.begin(), pp_terms.end());
    terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
    Ext z_h_zeta = b.sub_extension(zeta_pow_deg, one);
    for (int i = 0; i < NC; i++) {
      Ext vanishing = reduce_ext(b, terms, b.convert_to_ext(alphas[i]));
      std::vector<Ext> chunk(p.quotient.begin() + (size_t)i * Q, p.quotient.begin() + (size_t)(i + 1) * Q);
      Ext recombined = reduce_ext(b, chunk, zeta_pow_deg);
      b.connect_extension(vanishing, b.mul_extension(z_h_zeta, recombined));
    }
  }

  // ---- FRI (upstream fri/recursive_verifier.rs verify_fri_proof)
  // proof of work: the response has proof_of_work_bits leading zeros (assert_leading_zeros -> range check)
  b.range_check(pow_response, 64 - c.cfg.proof_of_work_bits);
  // precomputed reduced openings: sum_j alpha^j v_j per batch (batch 0: everything at zeta; batch 1: Z at g zeta)
  std::vector<Ext> batch0;
  for (auto* v : {&p.constants, &p.sigmas, &p.wires, &p.zs, &p.pps, &p.quotient}) batch0.insert(batch0.end(), v->begin(), v->end());
  const Ext reduced0 = reduce_ext(b, batch0, fri_alpha), reduced1 = reduce_ext(b, p.zs_next, fri_alpha);
  const Ext zeta_next = b.mul_const_extension(gl::root_of_unity(db), zeta);
  const Ext alpha_pow_b1 = b.exp_u64_extension(fri_alpha, (u64)NC);  // shift by the size of batch 1
  const std::vector<Hash>* caps[4] = {&cs_cap, &p.wires_cap, &p.zs_cap, &p.quotient_cap};

  for (int qi = 0; qi < c.cfg.num_query_rounds; qi++) {
    const ProofTargets::Query& q = p.queries[qi];
    std::vector<BoolTarget> all_bits = b.split_le(query_indices[qi], 64);           // low_bits(x, n_log, 64)
    std::vector<BoolTarget> x_bits(all_bits.begin(), all_bits.begin() + lde_bits);
    std::vector<BoolTarget> cap_bits(x_bits.end() - c.cfg.cap_height, x_bits.end());
    // "let cap_index = self.le_sum(x_index_bits[x_index_bits.len() - params.config.cap_height..].iter())"
    const Target cap_index = b.le_sum(cap_bits);
    const bool ra_cap = (1 << c.cfg.cap_height) == RA_VEC || c.cfg.cap_height == 0;
    const std::vector<Target> cap_onehot = ra_cap ? std::vector<Target>() : one_hot(b, cap_bits);
    for (int o = 0; o < 4; o++) verify_merkle_proof_to_cap(b, q.leaf[o], x_bits, cap_index, cap_onehot, *caps[o], q.path[o]);
    // subgroup_x = g * phi^rev(x_index): the point of the LDE coset this leaf sits at
    std::vector<BoolTarget> rev_bits(x_bits.rbegin(), x_bits.rend());
    Target subgroup_x = b.mul(b.constant(gl::GENERATOR), exp_from_bits_const_base(b, gl::root_of_unity(lde_bits), rev_bits));
    // fri_combine_initial: sum over batches of (reduced evals - reduced openings) / (x - point), alpha-shifted
    Ext old_eval;
    {
      Ext sx = b.convert_to_ext(subgroup_x);
      std::vector<Target> evals0;
      for (int o = 0; o < 4; o++) evals0.insert(evals0.end(), q.leaf[o].begin(), q.leaf[o].end());
      std::vector<Target> evals1(q.leaf[2].begin(), q.leaf[2].begin() + NC);
      Ext num0 = b.sub_extension(reduce_base(b, evals0, fri_alpha), reduced0);
      Ext sum = b.div_add_extension(num0, b.sub_extension(sx, zeta), b.zero_extension());
      sum = b.mul_extension(sum, alpha_pow_b1);
      Ext num1 = b.sub_extension(reduce_base(b, evals1, fri_alpha), reduced1);
      sum = b.div_add_extension(num1, b.sub_extension(sx, zeta_next), sum);
      // the batched polynomial was multiplied by X before the low-degree test (upstream PR #436)
      old_eval = b.mul_extension(sum, sx);
    }
    std::vector<BoolTarget> idx_bits = x_bits;
    for (size_t l = 0; l < c.fri_reduction_arity_bits.size(); l++) {
      const int ab = c.fri_reduction_arity_bits[l], arity = 1 << ab;
      const std::vector<Ext>& evals = q.step_evals[l];
      std::vector<BoolTarget> within(idx_bits.begin(), idx_bits.begin() + ab);
      std::vector<BoolTarget> coset_bits(idx_bits.begin() + ab, idx_bits.end());
      // consistency with the previous layer: evals[x_index_within_coset] == old_eval  (random_access_extension)
      if (arity == RA_VEC) {
        // "let x_index_within_coset = self.le_sum(..); let new_eval = self.random_access_extension(x_index_within_coset, evals.clone())"
        const Target within_index = b.le_sum(within);
        b.connect_extension(b.random_access_extension(within_index, evals), old_eval);
      } else {
        std::vector<Target> oh = one_hot(b, within);
        Ext sel = b.zero_extension();
        for (int k = 0; k < arity; k++) sel = b.mul_add_extension(b.convert_to_ext(oh[k]), evals[k], sel);
        b.connect_extension(sel, old_eval);
      }
      // compute_evaluation: interpolate the 2^ab values on the coset {s w^i} and evaluate at beta.
      {
        const u64 g = gl::root_of_unity(ab), g_inv = gl::inv(g);
        std::vector<BoolTarget> within_rev(within.rbegin(), within.rend());
        Target start = exp_from_bits_const_base(b, g_inv, within_rev);
        Target s = b.mul(start, subgroup_x);                                        // coset_start
        std::vector<Ext> ev(arity);
        for (int k = 0; k < arity; k++) ev[gl::bitrev((u32)k, ab)] = evals[k];   // reverse_index_bits
        const Ext beta = fri_betas[l];
        if (arity == CI_POINTS && c.cfg.max_quotient_degree_factor == 8) {
          // "let interpolation_gate = CosetInterpolationGate::with_max_degree(arity_bits, max_quotient_degree_factor);
          //  self.interpolate_coset(interpolation_gate, coset_start, &evals, beta)"
          old_eval = b.interpolate_coset(s, ev, beta);
        } else {
          //   P(beta) = (beta^m - s^m) / (m s^(m-1)) * sum_i v_i w^i / (beta - s w^i)   (barycentric form on arithmetic gates)
          Ext sum = b.zero_extension();
          u64 wi = 1;
          for (int i = 0; i < arity; i++) {
            Ext point = b.convert_to_ext(b.mul(b.constant(wi), s));
            Ext numer = b.mul_const_extension(wi, ev[i]);
            sum = b.div_add_extension(numer, b.sub_extension(beta, point), sum);
            wi = gl::mul(wi, g);
          }
          Target s_pow = b.exp_power_of_2(s, ab);                                    // s^m
          Ext z_beta = b.sub_extension(b.exp_power_of_2_extension(beta, ab), b.convert_to_ext(s_pow));
          // 1 / (m s^(m-1)) = s / (m s^m)
          Target scale = b.mul(s, b.inverse(b.mul(b.constant((u64)arity), s_pow)));
          old_eval = b.mul_extension(b.scalar_mul_ext(scale, z_beta), sum);
        }
      }
      std::vector<Target> flat;
      for (const Ext& e : evals) {
        flat.push_back(e[0]);
        flat.push_back(e[1]);
      }
      verify_merkle_proof_to_cap(b, flat, coset_bits, cap_index, cap_onehot, p.fri_caps[l], q.step_path[l]);
      subgroup_x = b.exp_power_of_2(subgroup_x, ab);
      idx_bits = coset_bits;
    }
    // final polynomial at the folded point
    Ext final_eval = reduce_ext(b, p.final_poly, b.convert_to_ext(subgroup_x));
    b.connect_extension(final_eval, old_eval);
  }
  return p;
}

}  // namespace

Circuit build_recursive_verifier(const Circuit& inner, const u64 digest[4], const std::vector<u64>& cs_cap, int n_proofs,
                                 bool expose_commitment) {
  if (n_proofs < 1 || n_proofs > 16) throw std::invalid_argument("recursive verifier: 1..16 inner proofs");
  if (cs_cap.size() != ((size_t)4 << inner.cfg.cap_height)) throw std::invalid_argument("recursive verifier: bad cap size");
  CircuitBuilder cb(inner.cfg);
  Hash dg;
  for (int i = 0; i < 4; i++) dg[i] = cb.constant(digest[i]);
  std::vector<Hash> cap(cs_cap.size() / 4);
  for (size_t k = 0; k < cap.size(); k++)
    for (int i = 0; i < 4; i++) cap[k][i] = cb.constant(cs_cap[4 * k + i]);
  std::vector<Target> ids;  // what identifies each inner proof to the outside
  for (int p = 0; p < n_proofs; p++) {
    ProofTargets pt = verify_one(cb, inner, dg, cap);
    if (!expose_commitment) continue;
    if (!pt.public_inputs.empty()) {
      ids.insert(ids.end(), pt.public_inputs.begin(), pt.public_inputs.end());  // an aggregate: its own commitment
    } else {
      // a leaf proof without public inputs: the hash of its wires commitment (binds the proof's witness)
      std::vector<Target> capw;
      for (const Hash& h : pt.wires_cap) capw.insert(capw.end(), h.begin(), h.end());
      Hash id = cb.hash_n_to_hash_no_pad(capw);
      ids.insert(ids.end(), id.begin(), id.end());
    }
  }
  if (expose_commitment) {
    // the aggregate's public inputs: one 4-word commitment to everything verified below it (a Poseidon tree over
    // the leaf proofs' identifiers when aggregation circuits are stacked)
    Hash root = ids.size() == 4 ? Hash{ids[0], ids[1], ids[2], ids[3]} : cb.hash_n_to_hash_no_pad(ids);
    cb.register_public_inputs({root[0], root[1], root[2], root[3]});
  }
  return cb.build();
}

}  // namespace p25
