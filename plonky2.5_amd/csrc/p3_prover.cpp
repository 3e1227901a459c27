// See p3_prover.h.
#include "p3_prover.h"
#include <array>
#include <functional>
#include <stdexcept>
#include <thread>
#include "poseidon2.h"

namespace p25 {
namespace {
using gl::E2;
typedef std::array<u64, 4> Digest;

void parallel_for(int threads, size_t n, const std::function<void(size_t, size_t)>& f) {
  if (threads <= 1 || n < 1024) {
    f(0, n);
    return;
  }
  std::vector<std::thread> th;
  size_t chunk = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    size_t b = t * chunk, e = std::min(n, b + chunk);
    if (b >= e) break;
    th.emplace_back([=, &f] { f(b, e); });
  }
  for (auto& x : th) x.join();
}

// natural-order radix-2 NTT with the primitive root `w` of order a.size()
void ntt(std::vector<u64>& a, u64 w) {
  const size_t n = a.size();
  unsigned lg = 0;
  while (((size_t)1 << lg) < n) lg++;
  for (size_t i = 0; i < n; i++) {
    size_t j = gl::bitrev((u32)i, lg);
    if (i < j) std::swap(a[i], a[j]);
  }
  std::vector<u64> tw(n / 2 ? n / 2 : 1);
  tw[0] = 1;
  for (size_t i = 1; i < n / 2; i++) tw[i] = gl::mul(tw[i - 1], w);
  for (size_t len = 1; len < n; len <<= 1) {
    size_t step = n / (2 * len);
    for (size_t s = 0; s < n; s += 2 * len)
      for (size_t j = 0; j < len; j++) {
        u64 u = a[s + j], v = gl::mul(a[s + j + len], tw[j * step]);
        a[s + j] = gl::add(u, v);
        a[s + j + len] = gl::sub(u, v);
      }
  }
}
void intt(std::vector<u64>& a) {
  unsigned lg = 0;
  while (((size_t)1 << lg) < a.size()) lg++;
  ntt(a, gl::inv(gl::root_of_unity(lg)));
  u64 ni = gl::inv((u64)a.size());
  for (auto& x : a) x = gl::mul(x, ni);
}
// values of the polynomial with coefficients c (len n) on shift*<w_{n 2^log_blowup}>, natural order
std::vector<u64> lde(const std::vector<u64>& c, u64 shift, int log_blowup) {
  const size_t n = c.size();
  std::vector<u64> v(n << log_blowup, 0);
  u64 p = 1;
  for (size_t k = 0; k < n; k++) {
    v[k] = gl::mul(c[k], p);
    p = gl::mul(p, shift);
  }
  unsigned lg = 0;
  while (((size_t)1 << lg) < v.size()) lg++;
  ntt(v, gl::root_of_unity(lg));
  return v;
}
E2 eval_ext(const std::vector<u64>& c, E2 x) {
  E2 acc = gl::e2(0);
  for (size_t i = c.size(); i-- > 0;) {
    acc = gl::mul(acc, x);
    acc.a = gl::add(acc.a, c[i]);
  }
  return acc;
}

// src/p3/challenger.rs:70-169
struct Challenger {
  u64 st[12] = {0};
  std::vector<u64> in, out;
  void duplex() {
    for (size_t i = 0; i < in.size(); i++) st[i] = in[i];
    in.clear();
    poseidon2::permute(st);
    out.assign(st, st + 12);
  }
  void observe(u64 x) {
    out.clear();
    in.push_back(x);
    if (in.size() == 12) duplex();
  }
  void observe_digest(const Digest& d) {
    for (u64 x : d) observe(x);
  }
  u64 sample() {
    if (!in.empty() || out.empty()) duplex();
    u64 v = out.back();
    out.pop_back();
    return v;
  }
  E2 sample_ext() {
    u64 a = sample();
    u64 b = sample();
    return E2{a, b};
  }
  u64 sample_bits(int bits) { return sample() & (((u64)1 << bits) - 1); }
};

// src/p3/commit.rs:23-60
Digest hash_row(const u64* row, size_t w) {
  u64 st[12] = {0};
  for (size_t off = 0; off < w; off += 4) {
    size_t m = std::min((size_t)4, w - off);
    for (size_t i = 0; i < m; i++) st[i] = row[off + i];
    poseidon2::permute(st);
  }
  return Digest{st[0], st[1], st[2], st[3]};
}
Digest compress(const Digest& l, const Digest& r) {
  u64 st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
  poseidon2::permute(st);
  return Digest{st[0], st[1], st[2], st[3]};
}
struct Tree {
  std::vector<std::vector<Digest>> levels;  // levels[0] leaves ... back() = root
  const Digest& root() const { return levels.back()[0]; }
  std::vector<Digest> prove(size_t index) const {
    std::vector<Digest> p;
    for (size_t k = 0; k + 1 < levels.size(); k++) p.push_back(levels[k][(index >> k) ^ 1]);
    return p;
  }
};
// rows: row-major matrix [height][width]
Tree commit(const std::vector<u64>& rows, size_t width, int threads) {
  const size_t h = rows.size() / width;
  Tree t;
  t.levels.emplace_back(h);
  parallel_for(threads, h, [&](size_t b, size_t e) {
    for (size_t i = b; i < e; i++) t.levels[0][i] = hash_row(&rows[i * width], width);
  });
  while (t.levels.back().size() > 1) {
    const auto& cur = t.levels.back();
    std::vector<Digest> nxt(cur.size() / 2);
    parallel_for(threads, nxt.size(), [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) nxt[i] = compress(cur[2 * i], cur[2 * i + 1]);
    });
    t.levels.push_back(std::move(nxt));
  }
  return t;
}
}  // namespace

namespace {
struct BaseOps {
  u64 cst(u64 v) { return v; }
  u64 add(u64 x, u64 y) { return gl::add(x, y); }
  u64 sub(u64 x, u64 y) { return gl::sub(x, y); }
  u64 mul(u64 x, u64 y) { return gl::mul(x, y); }
};
struct ExtOps {
  E2 cst(u64 v) { return gl::e2(v); }
  E2 add(E2 x, E2 y) { return gl::add(x, y); }
  E2 sub(E2 x, E2 y) { return gl::sub(x, y); }
  E2 mul(E2 x, E2 y) { return gl::mul(x, y); }
};
}  // namespace

std::vector<u64> p3_prove_fibonacci(const P3ProveParams& prm, P3Config& cfg) {
  if (prm.log_n < 1 || prm.log_n > 22) throw std::invalid_argument("p3_prove_fibonacci: unsupported parameters");
  // trace (src/p3/mod.rs:160-221): a, b, c = a + b; next a = b, next b = c; first row (1, 1)
  const size_t n = (size_t)1 << prm.log_n;
  std::vector<std::vector<u64>> col(3, std::vector<u64>(n));
  u64 a = 1, b = 1;
  for (size_t i = 0; i < n; i++) {
    u64 c = gl::add(a, b);
    col[0][i] = a;
    col[1][i] = b;
    col[2][i] = c;
    a = b;
    b = c;
  }
  return p3_prove_air(AirProgram::fibonacci(), col, prm, cfg);
}

std::vector<u64> p3_prove_air(const AirProgram& air, const std::vector<std::vector<u64>>& col, const P3ProveParams& prm,
                              P3Config& cfg) {
  // FriConfig.log_blowup (src/p3/mod.rs:242-246; verifier.rs uses config.log_blowup generically): the LDE domain is
  // 7*H_{n 2^B}, B = 1..4 (what p25_circuit_build_p3_verifier accepts); B = 1 is the reference's artifact, B = 2, 3 hold AIRs of
  // constraint degree up to 5 / 9 (four / eight chunks)
  if (prm.log_n < 1 || prm.log_n > 22 || prm.log_blowup < 1 || prm.log_blowup > 4 || prm.log_n + prm.log_blowup > 24 ||
      prm.num_queries < 1 || prm.pow_bits < 0 || prm.pow_bits > 30)
    throw std::invalid_argument("p3_prove: unsupported parameters");
  air.validate();
  const int k = prm.log_n, B = prm.log_blowup, L = k + B, T = prm.threads;
  const size_t n = (size_t)1 << k, N2 = n << B;      // N2: the LDE domain's size (2 n for the reference's log_blowup 1)
  const u64 w_n = gl::root_of_unity(k), w_2n = gl::root_of_unity(L);
  const int W = air.width;
  if ((int)col.size() != W) throw std::invalid_argument("p3_prove: trace width does not match the AIR");
  for (auto& c : col) {
    if (c.size() != n) throw std::invalid_argument("p3_prove: trace height does not match log_n");
    for (u64 v : c)
      if (v >= gl::P) throw std::invalid_argument("p3_prove: non-canonical trace value");
  }
  std::vector<std::vector<u64>> coef(col);
  for (auto& c : coef) intt(c);
  // LDE on 7*H_{2n}: natural order, then the committed matrix in bit-reversed row order
  std::vector<std::vector<u64>> lde_nat(W);
  for (int c = 0; c < W; c++) lde_nat[c] = lde(coef[c], gl::GENERATOR, B);
  std::vector<u64> trace_rows(N2 * W);
  for (size_t i = 0; i < N2; i++) {
    size_t r = gl::bitrev((u32)i, L);
    for (int c = 0; c < W; c++) trace_rows[i * W + c] = lde_nat[c][r];
  }
  Tree trace_tree = commit(trace_rows, W, T);

  Challenger ch;
  ch.observe_digest(trace_tree.root());
  const E2 alpha = ch.sample_ext();

  // Quotient chunks: 2^lqd of them, lqd = log2_ceil(max constraint degree - 1) with the selector's degree counted (uni-stark's
  // get_log_quotient_degree).  Degree <= 2 (every AIR the reference can verify: serde/proof.rs:41-48 holds one chunk): lqd = 0;
  // degree 3: lqd = 1; degree 4, 5: lqd = 2 (needs log_blowup >= 2).  The quotient domain is the disjoint coset 7*H_{n 2^lqd}
  // (verifier.rs:120-124): points x_j = 7 w^j = LDE natural index j * 2^(B - lqd), so the trace's LDE already holds the trace
  // there as long as lqd <= log_blowup.
  const int lqd = air.log_quotient_degree();
  if (lqd > B) throw std::invalid_argument("p3_prove: the AIR's constraint degree needs more quotient chunks than log_blowup holds");
  const size_t Q = (size_t)1 << lqd, nq = n << lqd;          // chunks, quotient-domain size
  const size_t lde_step = (size_t)1 << (B - lqd);              // LDE index step between quotient-domain points
  const size_t row_step = (size_t)1 << B;                      // ... and between a row and the next (x w_n)
  std::vector<u64> q0(nq), q1(nq);
  {
    const u64 g_inv = gl::inv(w_n);
    const u64 w_q = gl::root_of_unity(k + lqd);
    const u64 gn = gl::pow(gl::GENERATOR, n);
    // x^n - 1 on the coset: 7^n (w_q^j)^n - 1 takes 2^lqd values (constant for one chunk)
    std::vector<u64> zh(Q), zh_inv(Q);
    for (size_t r = 0; r < Q; r++) {
      zh[r] = gl::sub(gl::mul(gn, gl::pow(w_q, r * n)), 1);
      zh_inv[r] = gl::inv(zh[r]);
    }
    u64 x = gl::GENERATOR;
    BaseOps ops;
    std::vector<u64> loc(W), nxt(W);
    for (size_t j = 0; j < nq; j++, x = gl::mul(x, w_q)) {
      const size_t i0 = lde_step * j, i1 = (lde_step * j + row_step) % N2;   // the next ROW is x w_n = 2^B LDE points on
      for (int c = 0; c < W; c++) {
        loc[c] = lde_nat[c][i0];
        nxt[c] = lde_nat[c][i1];
      }
      // two_adic.rs:100-147: selectors at a point of the quotient coset
      const u64 zhx = zh[j & (Q - 1)];
      const u64 is_trans = gl::sub(x, g_inv);
      const u64 sel[4] = {0, gl::mul(zhx, gl::inv(gl::sub(x, 1))), gl::mul(zhx, gl::inv(is_trans)), is_trans};
      E2 acc = gl::e2(0);
      air.fold<u64>(loc, nxt, sel, ops, [&](u64 c) {  // VerifierConstraintFolder::assert_zero: acc = acc * alpha + c
        acc = gl::mul(acc, alpha);
        acc.a = gl::add(acc.a, c);
      });
      E2 q = gl::mul(acc, zh_inv[j & (Q - 1)]);
      q0[j] = q.a;
      q1[j] = q.b;
    }
  }
  // Chunk c = every Q-th value starting at c (TwoAdicMultiplicativeCoset::split_evals): the quotient on the coset
  // s_c H_n, s_c = 7 w_q^c (split_domains; p3_circuit.cpp quotient_chunks_domains).  Its two base columns are committed as
  // TwoAdicFriPcs::commit does: read as values on H_n -- the polynomial P_c(X) = q_c(s_c X) -- and extended to the coset
  // (7 / s_c) H_{2n}, i.e. q_c on 7*H_{2n}; all chunks are matrices of ONE Merkle tree (same height: rows concatenated).
  std::vector<std::vector<u64>> qcoef(2 * Q);     // P_c's coefficients, [2 c + component]
  std::vector<u64> s_c(Q);
  std::vector<u64> quot_rows(N2 * 2 * Q);
  {
    const u64 w_q = gl::root_of_unity(k + lqd);
    for (size_t c = 0; c < Q; c++) {
      s_c[c] = gl::mul(gl::GENERATOR, gl::pow(w_q, c));
      for (int comp = 0; comp < 2; comp++) {
        std::vector<u64> v(n);
        const std::vector<u64>& src = comp ? q1 : q0;
        for (size_t j = 0; j < n; j++) v[j] = src[j * Q + c];
        intt(v);
        std::vector<u64> l = lde(v, gl::mul(gl::GENERATOR, gl::inv(s_c[c])), B);
        for (size_t i = 0; i < N2; i++) quot_rows[i * 2 * Q + 2 * c + comp] = l[gl::bitrev((u32)i, L)];
        qcoef[2 * c + comp] = std::move(v);
      }
    }
  }
  Tree quot_tree = commit(quot_rows, (int)(2 * Q), T);
  ch.observe_digest(quot_tree.root());
  const E2 zeta = ch.sample_ext();
  const E2 zeta_next = gl::mul(zeta, w_n);

  // opened values
  std::vector<E2> t_local(W), t_next(W);
  for (int c = 0; c < W; c++) {
    t_local[c] = eval_ext(coef[c], zeta);
    t_next[c] = eval_ext(coef[c], zeta_next);
  }
  std::vector<E2> qz(2 * Q);                      // q_c's components at zeta = P_c at zeta / s_c
  for (size_t c = 0; c < Q; c++) {
    const E2 z_over_s = gl::mul(zeta, gl::inv(s_c[c]));
    for (int comp = 0; comp < 2; comp++) qz[2 * c + comp] = eval_ext(qcoef[2 * c + comp], z_over_s);
  }
  {  // self-check of the identity the verifier enforces (verifier.rs:169-239)
    E2 un = zeta;
    E2 z_h = gl::sub(gl::exp_pow2(un, k), gl::e2(1));
    E2 is_trans = gl::sub(un, gl::e2(gl::inv(w_n)));
    const E2 sel[4] = {gl::e2(0), gl::mul(z_h, gl::inv(gl::sub(un, gl::e2(1)))), gl::mul(z_h, gl::inv(is_trans)), is_trans};
    ExtOps ops;
    E2 acc = gl::e2(0);
    air.fold<E2>(t_local, t_next, sel, ops, [&](E2 c) { acc = gl::add(gl::mul(acc, alpha), c); });
    E2 lhs = gl::mul(acc, gl::inv(z_h));
    // quotient(zeta) = sum_c zps_c (q_c.a + X q_c.b),  zps_c = prod_{j != c} Z_{D_j}(zeta) / Z_{D_j}(s_c),  Z_{D_j}(x) = (x / s_j)^n - 1
    E2 rhs = gl::e2(0);
    for (size_t c = 0; c < Q; c++) {
      E2 zp = gl::e2(1);
      for (size_t j = 0; j < Q; j++) {
        if (j == c) continue;
        const u64 sj_inv = gl::inv(s_c[j]);
        E2 at_zeta = gl::sub(gl::exp_pow2(gl::mul(zeta, sj_inv), k), gl::e2(1));
        u64 at_first = gl::sub(gl::pow(gl::mul(s_c[c], sj_inv), n), 1);
        zp = gl::mul(zp, gl::mul(at_zeta, gl::inv(at_first)));
      }
      rhs = gl::add(rhs, gl::mul(zp, gl::add(qz[2 * c], gl::mul(qz[2 * c + 1], E2{0, 1}))));
    }
    if (!gl::eq(lhs, rhs)) throw std::logic_error("p3 prover: quotient identity does not hold");
  }

  // FRI input: reduced openings on the LDE domain, bit-reversed index (verifier.rs:296-338)
  const E2 fri_alpha = ch.sample_ext();
  std::vector<E2> folded(N2);
  {
    std::vector<E2> apow(2 * W + 2 * Q);
    apow[0] = gl::e2(1);
    for (size_t i = 1; i < apow.size(); i++) apow[i] = gl::mul(apow[i - 1], fri_alpha);
    parallel_for(T, N2, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) {
        size_t r = gl::bitrev((u32)i, L);
        u64 x = gl::mul(gl::GENERATOR, gl::pow(w_2n, r));
        E2 inv_z = gl::inv(gl::sub(gl::e2(x), zeta));
        E2 inv_zn = gl::inv(gl::sub(gl::e2(x), zeta_next));
        E2 acc = gl::e2(0);
        int t = 0;
        for (int c = 0; c < W; c++, t++)
          acc = gl::add(acc, gl::mul(apow[t], gl::mul(gl::sub(gl::e2(trace_rows[i * W + c]), t_local[c]), inv_z)));
        for (int c = 0; c < W; c++, t++)
          acc = gl::add(acc, gl::mul(apow[t], gl::mul(gl::sub(gl::e2(trace_rows[i * W + c]), t_next[c]), inv_zn)));
        for (size_t c = 0; c < 2 * Q; c++, t++)     // the chunk matrices in order, two columns each, all opened at zeta
          acc = gl::add(acc, gl::mul(apow[t], gl::mul(gl::sub(gl::e2(quot_rows[2 * Q * i + c]), qz[c]), inv_z)));
        folded[i] = acc;
      }
    });
  }
  // commit phase (verifier.rs:357-388 transcript, 441-516 fold)
  std::vector<Tree> fri_trees;
  std::vector<std::vector<E2>> fri_layers;
  {
    size_t m = N2;
    int lm = L;
    for (int round = 0; round < k; round++) {
      std::vector<u64> rows(m * 2);
      for (size_t i = 0; i < m; i++) {
        rows[2 * i] = folded[i].a;
        rows[2 * i + 1] = folded[i].b;
      }
      fri_layers.push_back(folded);
      fri_trees.push_back(commit(rows, 4, T));
      ch.observe_digest(fri_trees.back().root());
      const E2 beta = ch.sample_ext();
      const u64 w_m = gl::root_of_unity(lm);
      std::vector<E2> nxt(m / 2);
      parallel_for(T, m / 2, [&](size_t b, size_t e) {
        for (size_t j = b; j < e; j++) {
          u64 x = gl::pow(w_m, gl::bitrev((u32)(2 * j), lm));
          E2 e0 = folded[2 * j], e1 = folded[2 * j + 1];
          // evals[0] + (beta - xs[0]) * (evals[1] - evals[0]) / (xs[1] - xs[0]),  xs = (x, -x)
          E2 num = gl::mul(gl::sub(e1, e0), gl::sub(beta, gl::e2(x)));
          u64 den_inv = gl::inv(gl::sub(gl::neg(x), x));
          nxt[j] = gl::add(e0, gl::mul(num, den_inv));
        }
      });
      folded.swap(nxt);
      m >>= 1;
      lm--;
    }
    bool constant = folded.size() == ((size_t)1 << B);     // k folds leave 2^B values of a constant polynomial
    for (size_t i = 1; constant && i < folded.size(); i++) constant = gl::eq(folded[0], folded[i]);
    if (!constant) throw std::logic_error("p3 prover: FRI did not fold to a constant");
  }
  const E2 final_poly = folded[0];
  // proof of work (challenger.rs:159-168): observe the witness, sample_bits(pow_bits) must be 0
  u64 pow_witness = prm.pow_start;
  for (;; pow_witness++) {
    if (pow_witness >= gl::P) throw std::logic_error("p3 prover: no PoW witness");
    Challenger c2 = ch;
    c2.observe(pow_witness);
    if (c2.sample_bits(prm.pow_bits) == 0) break;
  }
  ch.observe(pow_witness);
  (void)ch.sample_bits(prm.pow_bits);
  std::vector<size_t> indices(prm.num_queries);
  for (auto& ix : indices) ix = (size_t)ch.sample_bits(L);

  // flatten in add_virtual_to order (proof.rs:357-373)
  cfg = P3Config();
  cfg.fri_config.log_blowup = prm.log_blowup;
  cfg.fri_config.num_queries = prm.num_queries;
  cfg.fri_config.proof_of_work_bits = prm.pow_bits;
  cfg.log_quotient_degree = lqd;
  cfg.log_trace_height = k;
  cfg.trace_width = W;
  cfg.opening_matrix_log_max_height = L;
  cfg.opening_proof_query_openings_opened_values_length = 2;
  cfg.degree_bits = k;
  std::vector<u64> out;
  auto push_d = [&](const Digest& d) { out.insert(out.end(), d.begin(), d.end()); };
  auto push_e = [&](E2 e) {
    out.push_back(e.a);
    out.push_back(e.b);
  };
  push_d(trace_tree.root());
  push_d(quot_tree.root());
  for (auto& e : t_local) push_e(e);
  for (auto& e : t_next) push_e(e);
  for (auto& e : qz) push_e(e);
  for (auto& t : fri_trees) push_d(t.root());
  for (size_t ix : indices) {
    size_t idx = ix;
    for (int round = 0; round < k; round++) {
      size_t sib = idx ^ 1, pair = idx >> 1;
      push_e(fri_layers[round][sib]);
      for (auto& d : fri_trees[round].prove(pair)) push_d(d);
      idx = pair;
    }
  }
  push_e(final_poly);
  out.push_back(pow_witness);
  for (size_t ix : indices) {
    for (int c = 0; c < W; c++) out.push_back(trace_rows[ix * W + c]);
    for (auto& d : trace_tree.prove(ix)) push_d(d);
    for (size_t c = 0; c < 2 * Q; c++) out.push_back(quot_rows[2 * Q * ix + c]);
    for (auto& d : quot_tree.prove(ix)) push_d(d);
  }
  if (out.size() != cfg.num_inputs()) throw std::logic_error("p3 prover: flattened size mismatch");
  return out;
}

std::string p3_inputs_to_json(const std::vector<u64>& in, const P3Config& cfg) {
  if (in.size() != cfg.num_inputs()) throw std::invalid_argument("input vector does not match the shape");
  size_t pos = 0;
  std::string s;
  auto fe = [&]() { s += "{\"value\":" + std::to_string(in[pos++]) + "}"; };
  auto arr = [&](size_t n) {
    s += '[';
    for (size_t i = 0; i < n; i++) {
      if (i) s += ',';
      fe();
    }
    s += ']';
  };
  auto val = [&](size_t n) {
    s += "{\"value\":";
    arr(n);
    s += '}';
  };
  auto list = [&](size_t n, const std::function<void()>& f) {
    s += '[';
    for (size_t i = 0; i < n; i++) {
      if (i) s += ',';
      f();
    }
    s += ']';
  };
  const int k = cfg.log_trace_height;
  s += "{\"commitments\":{\"trace\":";
  val(4);
  s += ",\"quotient_chunks\":";
  val(4);
  s += "},\"opened_values\":{\"trace_local\":";
  list(cfg.trace_width, [&] { val(2); });
  s += ",\"trace_next\":";
  list(cfg.trace_width, [&] { val(2); });
  s += ",\"quotient_chunks\":";
  list((size_t)1 << cfg.log_quotient_degree, [&] { list(2, [&] { val(2); }); });
  s += "},\"opening_proof\":{\"fri_proof\":{\"commit_phase_commits\":";
  list(k, [&] { val(4); });
  s += ",\"query_proofs\":";
  list(cfg.fri_config.num_queries, [&] {
    s += "{\"commit_phase_openings\":";
    int i = 0;
    list(k, [&] {
      s += "{\"sibling_value\":";
      val(2);
      s += ",\"opening_proof\":";
      list(cfg.opening_matrix_log_max_height - 1 - i, [&] { arr(4); });
      s += '}';
      i++;
    });
    s += '}';
  });
  s += ",\"final_poly\":";
  val(2);
  s += ",\"pow_witness\":";
  fe();
  s += "},\"query_openings\":";
  list(cfg.fri_config.num_queries, [&] {
    int widths[2] = {cfg.trace_width, cfg.opening_proof_query_openings_opened_values_length};
    int b = 0;
    list(2, [&] {
      s += "{\"opened_values\":";
      list(b == 0 ? 1 : (size_t)1 << cfg.log_quotient_degree, [&] { arr(widths[b]); });
      s += ",\"opening_proof\":";
      list(cfg.opening_matrix_log_max_height, [&] { arr(4); });
      s += '}';
      b++;
    });
  });
  s += "},\"degree_bits\":" + std::to_string(cfg.degree_bits) + "}";
  if (pos != in.size()) throw std::logic_error("p3 json: size mismatch");
  return s;
}

}  // namespace p25
