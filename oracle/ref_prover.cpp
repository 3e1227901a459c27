// ORACLE -- test infrastructure only (see ref_prover.h).
#include "ref_prover.h"
#include "ref_hash_x8.h"
#include <string.h>
#include <chrono>
#include <functional>
#include <thread>
#include "ref_fft.h"
#include "ref_gates.h"

// Threading model of the CPU baseline: a persistent worker pool (no thread creation per loop) and a
// per-calling-thread width, so that N independent proofs can run on N cores, one thread each
// (ref_set_thread_local_threads(1) in every proving thread), or one proof can be spread over T cores.
#include <atomic>
#include <condition_variable>
#include <mutex>
static int g_threads = 1;
static thread_local int tl_threads = 0;       // 0 = use the process-wide setting
static thread_local bool tl_in_worker = false;
void ref_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
// The tuned cpu_baseline leg (bench.py): Merkle trees hashed eight leaves / eight parents at a time on AVX-512
// (ref_hash_x8.cpp).  Off by default: the checker runs the scalar code.  Same digests either way.
static bool g_tuned = false;
int ref_set_tuned(int on) {
  g_tuned = on && ref_x8_available();
  ref_fft_set_tuned(g_tuned);
  return g_tuned ? 1 : 0;
}
void ref_set_thread_local_threads(int n) { tl_threads = n < 0 ? 0 : n; }

namespace {
struct Job {
  const std::function<void(size_t, size_t)>* f;
  size_t n, chunk;
  std::atomic<size_t> next{0};
  std::atomic<size_t> done{0};
  size_t total_chunks;
  std::mutex m;
  std::condition_variable cv;
};
class Pool {
 public:
  void run(int T, size_t n, const std::function<void(size_t, size_t)>& f) {
    Job job;
    job.f = &f;
    job.n = n;
    job.chunk = (n + T - 1) / T;
    job.total_chunks = (n + job.chunk - 1) / job.chunk;
    {
      std::lock_guard<std::mutex> lk(m_);
      // several proofs may be running their loops at once (one group of T threads each): enough helpers for all
      demand_ += T - 1;
      while ((int)workers_.size() < demand_ && workers_.size() < 1024) workers_.emplace_back([this] { worker(); });
      for (int i = 0; i < T - 1; i++) q_.push_back(&job);
    }
    cv_.notify_all();
    work(job);
    std::unique_lock<std::mutex> lk(job.m);
    job.cv.wait(lk, [&] { return job.done.load() == job.total_chunks; });
    // drop queue entries that no worker picked up (all chunks were already taken)
    lk.unlock();
    {
      std::lock_guard<std::mutex> lk2(m_);
      for (auto it = q_.begin(); it != q_.end();) it = (*it == &job) ? q_.erase(it) : it + 1;
      demand_ -= T - 1;
    }
    // a worker that popped the job but is still inside work()/notify must be waited for (job is on this stack)
    for (;;) {
      {
        std::lock_guard<std::mutex> lk2(m_);
        if (!job_refs(&job)) break;
      }
      std::this_thread::yield();
    }
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  std::vector<std::thread> workers_;
  std::vector<Job*> q_;
  std::vector<Job*> active_;
  int demand_ = 0;  // helpers wanted by the loops running right now
  bool job_refs(Job* j) {
    for (Job* a : active_)
      if (a == j) return true;
    return false;
  }
  static void work(Job& job) {
    for (;;) {
      size_t c = job.next.fetch_add(1);
      if (c >= job.total_chunks) return;
      size_t b = c * job.chunk, e = b + job.chunk < job.n ? b + job.chunk : job.n;
      (*job.f)(b, e);
      if (job.done.fetch_add(1) + 1 == job.total_chunks) {
        std::lock_guard<std::mutex> lk(job.m);
        job.cv.notify_all();
      }
    }
  }
  void worker() {
    tl_in_worker = true;
    for (;;) {
      Job* j;
      {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty(); });
        j = q_.back();
        q_.pop_back();
        active_.push_back(j);
      }
      work(*j);
      {
        std::lock_guard<std::mutex> lk(m_);
        for (auto it = active_.begin(); it != active_.end(); ++it)
          if (*it == j) {
            active_.erase(it);
            break;
          }
      }
    }
  }
};
Pool& pool() {
  static Pool* p = new Pool();  // leaked on purpose: workers are detached-for-life daemon threads
  return *p;
}
}  // namespace

static void parallel_for(size_t n, const std::function<void(size_t, size_t)>& f) {
  int T = tl_threads ? tl_threads : g_threads;
  if (T <= 1 || n < 2 || tl_in_worker) {
    f(0, n);
    return;
  }
  if ((size_t)T > n) T = (int)n;
  pool().run(T, n, f);
}
static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ---------------------------------------------------------------- commitments (App. A.3)
static RMerkleTree merkle_levels(std::vector<RHash> cur, unsigned cap_height) {
  RMerkleTree t;
  t.cap_height = cap_height;
  t.levels.push_back(cur);
  while (cur.size() > ((size_t)1 << cap_height)) {
    std::vector<RHash> nxt(cur.size() / 2);
    if (g_tuned && nxt.size() % 8 == 0) {
      parallel_for(nxt.size() / 8, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) ref_two_to_one_x8(&cur[16 * i], &nxt[8 * i]);
      });
    } else {
      parallel_for(nxt.size(), [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) nxt[i] = ref_two_to_one(cur[2 * i], cur[2 * i + 1]);
      });
    }
    t.levels.push_back(nxt);
    cur.swap(nxt);
  }
  return t;
}
static RMerkleTree merkle_parallel(const std::vector<std::vector<u64>>& leaves, unsigned cap_height) {
  std::vector<RHash> cur(leaves.size());
  parallel_for(leaves.size(), [&](size_t b, size_t e) {
    for (size_t i = b; i < e; i++) cur[i] = ref_hash_or_noop(leaves[i].data(), leaves[i].size());
  });
  return merkle_levels(std::move(cur), cap_height);
}
static RMerkleTree merkle_parallel_flat(const u64* leaves, size_t n_leaves, size_t width, unsigned cap_height) {
  std::vector<RHash> cur(n_leaves);
  if (g_tuned && width > 4 && n_leaves % 8 == 0) {
    parallel_for(n_leaves / 8, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) ref_hash_rows_x8(leaves, width, 8 * i, &cur[8 * i]);
    });
  } else {
    parallel_for(n_leaves, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) cur[i] = ref_hash_or_noop(leaves + i * width, width);
    });
  }
  return merkle_levels(std::move(cur), cap_height);
}

RPolyBatch ref_commit_coeffs(const std::vector<std::vector<u64>>& coeffs, int rate_bits, int cap_height) {
  RPolyBatch b;
  b.n_polys = coeffs.size();
  b.rate_bits = rate_bits;
  const size_t n = coeffs[0].size();
  b.log_n = 0;
  while (((size_t)1 << b.log_n) < n) b.log_n++;
  b.coeffs = coeffs;
  const size_t big = n << rate_bits;
  // One flat row-major matrix (a leaf = one contiguous row), filled eight columns at a time so that every 64-byte
  // line of it is written once, in address order: the LDE comes out of the decimation-in-frequency transform already
  // at bit-reversed index.  (A vector of 2^19 little vectors scattered one word at a time made this the slowest part
  // of the CPU baseline and stopped it scaling across cores.)
  b.leaves_flat.assign(big * b.n_polys, 0);
  const size_t np = b.n_polys, blocks = (np + 7) / 8;
  parallel_for(blocks, [&](size_t bb, size_t be) {
    std::vector<std::vector<u64>> cols;
    for (size_t blk = bb; blk < be; blk++) {
      const size_t p0 = blk * 8, p1 = p0 + 8 < np ? p0 + 8 : np;
      cols.clear();
      for (size_t p = p0; p < p1; p++) cols.push_back(ref_lde_values_bitrev(coeffs[p], rate_bits, 7));
      for (size_t r = 0; r < big; r++) {
        u64* row = b.leaves_flat.data() + r * np + p0;
        for (size_t k = 0; k < cols.size(); k++) row[k] = cols[k][r];
      }
    }
  });
  b.tree = merkle_parallel_flat(b.leaves_flat.data(), big, b.n_polys, cap_height);
  return b;
}
RPolyBatch ref_commit_values(const std::vector<std::vector<u64>>& values, int rate_bits, int cap_height) {
  std::vector<std::vector<u64>> coeffs(values);
  parallel_for(coeffs.size(), [&](size_t b, size_t e) {
    for (size_t p = b; p < e; p++) ref_ifft(coeffs[p]);
  });
  return ref_commit_coeffs(coeffs, rate_bits, cap_height);
}

RPrecomputed ref_precompute(const RCircuit& c) {
  RPrecomputed pre;
  pre.constants_sigmas = ref_commit_values(c.constants_sigmas, c.rate_bits, c.cap_height);
  // circuit_digest = H(cap || H_pad(domain_separator = []) || [degree_bits])  (App. A.1)
  std::vector<u64> parts;
  for (auto& h : pre.constants_sigmas.tree.cap())
    for (int i = 0; i < 4; i++) parts.push_back(h.e[i]);
  u64 pad[8] = {1, 0, 0, 0, 0, 0, 0, 1};  // pad10*1 of the empty message to a multiple of the rate
  RHash ds = ref_hash_no_pad(pad, 8);
  for (int i = 0; i < 4; i++) parts.push_back(ds.e[i]);
  parts.push_back((u64)c.degree_bits);
  pre.circuit_digest = ref_hash_no_pad(parts.data(), parts.size());
  return pre;
}

// ---------------------------------------------------------------- proof layout
static int n_oracles() { return 4; }
static void oracle_widths(const RCircuit& c, int w[4]) {
  w[0] = c.num_cs();
  w[1] = c.num_wires;
  w[2] = c.num_challenges * (1 + c.num_partial_products);
  w[3] = c.num_challenges * c.quotient_degree_factor;
}
static int final_poly_len(const RCircuit& c) {
  int db = c.degree_bits;
  for (int a : c.arity_bits) db -= a;
  return 1 << db;
}
size_t ref_proof_words(const RCircuit& c) {
  const size_t capw = (size_t)4 << c.cap_height;
  int w[4];
  oracle_widths(c, w);
  size_t total = 3 * capw;
  total += 2 * (size_t)(c.num_constants_total() + c.num_routed + c.num_wires + 2 * c.num_challenges +
                        c.num_challenges * c.num_partial_products + w[3]);
  total += c.arity_bits.size() * capw;
  const int lde_bits = c.degree_bits + c.rate_bits;
  size_t per_q = 0;
  for (int o = 0; o < 4; o++) per_q += w[o] + 4 * (size_t)(lde_bits - c.cap_height);
  int bits = lde_bits;
  for (int a : c.arity_bits) {
    bits -= a;
    per_q += 2 * ((size_t)1 << a) + 4 * (size_t)(bits - c.cap_height);
  }
  total += per_q * c.num_queries;
  total += 2 * (size_t)final_poly_len(c) + 1;
  total += c.public_inputs.size();
  return total;
}
std::vector<u64> ref_proof_flatten(const RCircuit& c, const RProof& p) {
  std::vector<u64> o;
  auto cap = [&](const std::vector<RHash>& v) {
    for (auto& h : v) o.insert(o.end(), h.e, h.e + 4);
  };
  auto exts = [&](const std::vector<RE2>& v) {
    for (auto& e : v) {
      o.push_back(e.a);
      o.push_back(e.b);
    }
  };
  cap(p.wires_cap); cap(p.zs_cap); cap(p.quotient_cap);
  exts(p.constants); exts(p.sigmas); exts(p.wires); exts(p.zs); exts(p.zs_next); exts(p.pps); exts(p.quotient);
  for (auto& fc : p.fri_caps) cap(fc);
  for (auto& q : p.queries) {
    for (size_t k = 0; k < q.initial_leaf.size(); k++) {
      o.insert(o.end(), q.initial_leaf[k].begin(), q.initial_leaf[k].end());
      cap(q.initial_path[k]);
    }
    for (size_t k = 0; k < q.step_evals.size(); k++) {
      exts(q.step_evals[k]);
      cap(q.step_path[k]);
    }
  }
  exts(p.final_poly);
  o.push_back(p.pow_witness);
  o.insert(o.end(), p.public_inputs.begin(), p.public_inputs.end());
  return o;
}
RProof ref_proof_unflatten(const RCircuit& c, const u64* w) {
  RProof p;
  size_t off = 0;
  auto cap = [&](std::vector<RHash>& v, size_t n) {
    v.resize(n);
    for (auto& h : v) {
      memcpy(h.e, w + off, 32);
      off += 4;
    }
  };
  auto exts = [&](std::vector<RE2>& v, size_t n) {
    v.resize(n);
    for (auto& e : v) {
      e.a = w[off++];
      e.b = w[off++];
    }
  };
  const size_t capn = (size_t)1 << c.cap_height;
  int ow[4];
  oracle_widths(c, ow);
  cap(p.wires_cap, capn); cap(p.zs_cap, capn); cap(p.quotient_cap, capn);
  exts(p.constants, c.num_constants_total()); exts(p.sigmas, c.num_routed); exts(p.wires, c.num_wires);
  exts(p.zs, c.num_challenges); exts(p.zs_next, c.num_challenges);
  exts(p.pps, (size_t)c.num_challenges * c.num_partial_products); exts(p.quotient, ow[3]);
  p.fri_caps.resize(c.arity_bits.size());
  for (auto& fc : p.fri_caps) cap(fc, capn);
  const int lde_bits = c.degree_bits + c.rate_bits;
  p.queries.resize(c.num_queries);
  for (auto& q : p.queries) {
    q.initial_leaf.resize(4);
    q.initial_path.resize(4);
    for (int k = 0; k < 4; k++) {
      q.initial_leaf[k].assign(w + off, w + off + ow[k]);
      off += ow[k];
      cap(q.initial_path[k], lde_bits - c.cap_height);
    }
    int bits = lde_bits;
    q.step_evals.resize(c.arity_bits.size());
    q.step_path.resize(c.arity_bits.size());
    for (size_t k = 0; k < c.arity_bits.size(); k++) {
      bits -= c.arity_bits[k];
      exts(q.step_evals[k], (size_t)1 << c.arity_bits[k]);
      cap(q.step_path[k], bits - c.cap_height);
    }
  }
  exts(p.final_poly, final_poly_len(c));
  p.pow_witness = w[off++];
  p.public_inputs.assign(w + off, w + off + c.public_inputs.size());
  return p;
}

// ---------------------------------------------------------------- helpers
static RE2 eval_poly_ext(const std::vector<u64>& coeffs, RE2 x) {
  RE2 acc = re(0);
  for (size_t i = coeffs.size(); i-- > 0;) acc = re_add(re_mul(acc, x), re(coeffs[i]));
  return acc;
}
static RE2 eval_ext_poly(const std::vector<RE2>& coeffs, RE2 x) {
  RE2 acc = re(0);
  for (size_t i = coeffs.size(); i-- > 0;) acc = re_add(re_mul(acc, x), coeffs[i]);
  return acc;
}
static void batch_inverse(std::vector<u64>& v) {
  std::vector<u64> pre(v.size());
  u64 acc = 1;
  for (size_t i = 0; i < v.size(); i++) {
    pre[i] = acc;
    acc = rf_mul(acc, v[i]);
  }
  u64 inv = rf_inv(acc);
  for (size_t i = v.size(); i-- > 0;) {
    u64 t = rf_mul(inv, pre[i]);
    inv = rf_mul(inv, v[i]);
    v[i] = t;
  }
}
static std::vector<u64> flatten_exts(const std::vector<RE2>& v) {
  std::vector<u64> o;
  for (auto& e : v) {
    o.push_back(e.a);
    o.push_back(e.b);
  }
  return o;
}
static int leading_zeros64(u64 x) { return x ? __builtin_clzll(x) : 64; }

// ---------------------------------------------------------------- partial products and Z (App. A.5)
// upstream prover.rs `wires_permutation_partial_products_and_zs`: rows 0..NC = Z, then NC*NP partial products
std::vector<std::vector<u64>> ref_partial_products(const RCircuit& c, const std::vector<std::vector<u64>>& wires_values,
                                                   const std::vector<u64>& betas, const std::vector<u64>& gammas) {
  const size_t n = c.n();
  const int NC = c.num_challenges, RW = c.num_routed, NP = c.num_partial_products, Q = c.quotient_degree_factor;
  const int n_consts = c.num_constants_total();
  struct { const std::vector<std::vector<u64>>& wires; } wr{wires_values};
  std::vector<u64> subgroup(n);
  {
    u64 g = rf_root_of_unity(c.degree_bits), x = 1;
    for (auto& s : subgroup) {
      s = x;
      x = rf_mul(x, g);
    }
  }
  std::vector<std::vector<u64>> zs_pp(NC * (1 + NP), std::vector<u64>(n));
  for (int ci = 0; ci < NC; ci++) {
    std::vector<std::vector<u64>> chunk_products(n, std::vector<u64>(NP + 1));
    parallel_for(n, [&](size_t rb, size_t re_) {
      std::vector<u64> num(RW), den(RW);
      for (size_t r = rb; r < re_; r++) {
        for (int j = 0; j < RW; j++) {
          u64 wv = wr.wires[j][r];
          u64 s_id = rf_mul(c.k_is[j], subgroup[r]);
          num[j] = rf_add(rf_add(wv, rf_mul(betas[ci], s_id)), gammas[ci]);
          den[j] = rf_add(rf_add(wv, rf_mul(betas[ci], c.constants_sigmas[n_consts + j][r])), gammas[ci]);
        }
        batch_inverse(den);
        for (int k = 0; k <= NP; k++) {
          u64 p = 1;
          for (int j = k * Q; j < (k + 1) * Q && j < RW; j++) p = rf_mul(p, rf_mul(num[j], den[j]));
          chunk_products[r][k] = p;
        }
      }
    });
    u64 z_x = 1;
    for (size_t r = 0; r < n; r++) {
      u64 acc = z_x;
      u64 zrow = z_x;
      for (int k = 0; k <= NP; k++) {
        acc = rf_mul(acc, chunk_products[r][k]);
        if (k < NP)
          zs_pp[NC + ci * NP + k][r] = acc;
        else
          z_x = acc;  // Z(g x) becomes the next row's Z(x)
      }
      zs_pp[ci][r] = zrow;
    }
  }
  return zs_pp;
}

// ---------------------------------------------------------------- quotient polynomials (App. A.6)
// upstream prover.rs `compute_quotient_polys` + "split up quotient polys": NC * Q coefficient vectors of length n
std::vector<std::vector<u64>> ref_quotient_chunks(const RCircuit& c, const RPolyBatch& constants_sigmas,
                                                  const RPolyBatch& wires, const RPolyBatch& zs_batch,
                                                  const std::vector<u64>& betas, const std::vector<u64>& gammas,
                                                  const std::vector<u64>& alphas, const u64* pih) {
  const size_t n = c.n();
  const int NC = c.num_challenges, RW = c.num_routed, NP = c.num_partial_products, Q = c.quotient_degree_factor;
  const int lde_bits = c.degree_bits + c.rate_bits;
  const size_t big = (size_t)1 << lde_bits;
  const int n_consts = c.num_constants_total();
  struct { const RPolyBatch& constants_sigmas; } pre{constants_sigmas};
  std::vector<std::vector<u64>> qvals(NC, std::vector<u64>(big));
  {
    const u64 w_big = rf_root_of_unity(lde_bits);
    const u64 g_pow_n = rf_pow(7, n);
    const u64 w_rate = rf_root_of_unity(c.rate_bits);
    std::vector<u64> zh(1 << c.rate_bits), zh_inv(1 << c.rate_bits);
    for (int k = 0; k < (1 << c.rate_bits); k++) {
      zh[k] = rf_sub(rf_mul(g_pow_n, rf_pow(w_rate, k)), 1);
      zh_inv[k] = rf_inv(zh[k]);
    }
    const u64 n_field = (u64)n;
    const size_t next_step = (size_t)1 << c.rate_bits;
    if (g_tuned && big % 8 == 0 && NC <= 8) {   // the tuned leg: eight points per pass (ref_quotient_x8.cpp), same values
      parallel_for(big / 8, [&](size_t gb, size_t ge) {
        u64 xs[8], l0s[8], outv[8][8];
        u64 x = rf_mul(7, rf_pow(w_big, gb * 8));
        for (size_t g = gb; g < ge; g++) {
          const size_t i0 = g * 8;
          for (int j = 0; j < 8; j++, x = rf_mul(x, w_big)) {
            xs[j] = x;
            l0s[j] = rf_mul(zh[(i0 + j) % (1 << c.rate_bits)], rf_inv(rf_mul(n_field, rf_sub(x, 1))));
          }
          ref_vanishing_points_x8(c, pre.constants_sigmas, wires, zs_batch, betas.data(), gammas.data(), alphas.data(), pih, i0,
                                  xs, l0s, outv);
          for (int k = 0; k < NC; k++)
            for (int j = 0; j < 8; j++) qvals[k][i0 + j] = rf_mul(outv[k][j], zh_inv[(i0 + j) % (1 << c.rate_bits)]);
        }
      });
    } else
    parallel_for(big, [&](size_t ib, size_t ie) {
      std::vector<FB> consts(n_consts), sig(RW), wv(c.num_wires), z(NC), zn(NC), pp(NC * NP);
      FB outv[8];
      u64 x = rf_mul(7, rf_pow(w_big, ib));
      for (size_t i = ib; i < ie; i++, x = rf_mul(x, w_big)) {
        size_t pos = rbits(i, lde_bits), pos_next = rbits((i + next_step) % big, lde_bits);
        const u64* cs = pre.constants_sigmas.leaf(pos);
        for (int k = 0; k < n_consts; k++) consts[k] = FB{cs[k]};
        for (int k = 0; k < RW; k++) sig[k] = FB{cs[n_consts + k]};
        const u64* wl = wires.leaf(pos);
        for (int k = 0; k < c.num_wires; k++) wv[k] = FB{wl[k]};
        const u64* zl = zs_batch.leaf(pos);
        const u64* znl = zs_batch.leaf(pos_next);
        for (int k = 0; k < NC; k++) {
          z[k] = FB{zl[k]};
          zn[k] = FB{znl[k]};
        }
        for (int k = 0; k < NC * NP; k++) pp[k] = FB{zl[NC + k]};
        u64 zhx = zh[i % (1 << c.rate_bits)];
        u64 l0 = rf_mul(zhx, rf_inv(rf_mul(n_field, rf_sub(x, 1))));
        ref_eval_vanishing<FB>(c, FB{x}, FB{l0}, consts.data(), sig.data(), wv.data(), z.data(), zn.data(),
                               pp.data(), betas.data(), gammas.data(), alphas.data(), pih, outv);
        for (int k = 0; k < NC; k++) qvals[k][i] = rf_mul(outv[k].v, zh_inv[i % (1 << c.rate_bits)]);
      }
    });
  }
  std::vector<std::vector<u64>> qchunks;
  for (int k = 0; k < NC; k++) {
    ref_coset_ifft(qvals[k], 7);
    for (int j = 0; j < Q; j++) qchunks.emplace_back(qvals[k].begin() + j * n, qvals[k].begin() + (j + 1) * n);
  }
  return qchunks;
}

// ---------------------------------------------------------------- FRI commit phase, PoW, queries (App. A.8)
// upstream fri/prover.rs `fri_proof`: given the batched polynomial (n extension coefficients) and the
// transcript, fills fri_caps, final_poly, pow_witness and the query rounds of `out`.  `oracles` (the four
// initial polynomial batches) may be null: then only the FRI layers are opened.
int ref_fri_prove(const RFriParams& fp, const std::vector<RE2>& final_poly, RChallenger& ch,
                  const RPolyBatch* const* oracles, RProof& out, std::vector<size_t>* indices_out, std::string* msg) {
  const size_t big = final_poly.size() << fp.rate_bits;
  // commit phase (App. A.8)
  std::vector<RE2> coeffs(final_poly);
  coeffs.resize(big, re(0));
  std::vector<RE2> values(coeffs);
  ref_coset_fft_ext(values, 7);
  u64 shift = 7;
  std::vector<RMerkleTree> fri_trees;
  std::vector<std::vector<std::vector<u64>>> fri_leaves;
  std::vector<RE2> fri_betas;
  for (int arity_bits : fp.arity_bits) {
    const size_t arity = (size_t)1 << arity_bits;
    const size_t len = values.size();
    unsigned lg = 0;
    while (((size_t)1 << lg) < len) lg++;
    std::vector<RE2> br(len);
    for (size_t i = 0; i < len; i++) br[rbits(i, lg)] = values[i];
    std::vector<std::vector<u64>> leaves(len / arity);
    for (size_t l = 0; l < leaves.size(); l++)
      for (size_t k = 0; k < arity; k++) {
        leaves[l].push_back(br[l * arity + k].a);
        leaves[l].push_back(br[l * arity + k].b);
      }
    RMerkleTree tree = merkle_parallel(leaves, fp.cap_height);
    ch.observe_cap(tree.cap());
    out.fri_caps.push_back(tree.cap());
    fri_trees.push_back(tree);
    fri_leaves.push_back(leaves);
    RE2 beta = ch.ext_challenge();
    fri_betas.push_back(beta);
    std::vector<RE2> folded(coeffs.size() / arity);
    for (size_t i = 0; i < folded.size(); i++) {
      RE2 s = re(0);
      for (size_t k = arity; k-- > 0;) s = re_add(re_mul(s, beta), coeffs[i * arity + k]);
      folded[i] = s;
    }
    coeffs.swap(folded);
    shift = rf_pow(shift, arity);
    values = coeffs;
    ref_coset_fft_ext(values, shift);
  }
  coeffs.resize(coeffs.size() >> fp.rate_bits);
  out.final_poly = coeffs;
  for (auto& e : coeffs) ch.observe_ext(e);

  // proof of work: smallest witness whose response has >= pow_bits leading zeros
  {
    u64 st[12];
    memcpy(st, ch.state, sizeof(st));
    size_t pos = ch.in.size();
    for (size_t i = 0; i < pos; i++) st[i] = ch.in[i];
    u64 cand = 0;
    for (;; cand++) {
      u64 s2[12];
      memcpy(s2, st, sizeof(st));
      s2[pos] = cand;
      ref_poseidon(s2);
      if (leading_zeros64(s2[7]) >= fp.pow_bits) break;
    }
    out.pow_witness = cand;
    ch.observe(cand);
    u64 resp = ch.challenge();
    if (leading_zeros64(resp) < fp.pow_bits) {
      if (msg) *msg = "internal: PoW response mismatch";
      return 7;
    }
  }
  // query rounds
  out.queries.resize(fp.num_queries);
  for (auto& q : out.queries) {
    size_t x_index = (size_t)(ch.challenge() % big);
    if (indices_out) indices_out->push_back(x_index);
    for (int o = 0; oracles && o < 4; o++) {
      q.initial_leaf.push_back(std::vector<u64>(oracles[o]->leaf(x_index), oracles[o]->leaf(x_index) + oracles[o]->n_polys));
      q.initial_path.push_back(oracles[o]->tree.prove(x_index));
    }
    for (size_t l = 0; l < fp.arity_bits.size(); l++) {
      size_t ci = x_index >> fp.arity_bits[l];
      std::vector<RE2> evals;
      for (size_t k = 0; k < fri_leaves[l][ci].size(); k += 2) evals.push_back(RE2{fri_leaves[l][ci][k], fri_leaves[l][ci][k + 1]});
      q.step_evals.push_back(evals);
      q.step_path.push_back(fri_trees[l].prove(ci));
      x_index = ci;
    }
  }
  return 0;
}

// ---------------------------------------------------------------- prover (App. A.0-A.8)
int ref_prove(const RCircuit& c, const RPrecomputed& pre, const u64* inputs, u64 seed, RProof& out, RTimings* tm,
              std::string* msg, const u64* filler) {
  RTimings T;
  double t_start = now_s(), t0 = t_start;
  const size_t n = c.n();
  const int NC = c.num_challenges;
  const int n_consts = c.num_constants_total();

  RWitnessResult wr = ref_generate_witness(c, inputs, seed, filler);
  if (wr.status) {
    if (msg) *msg = wr.message;
    return wr.status;
  }
  T.witness = now_s() - t0; t0 = now_s();
  // "let public_inputs_hash = C::InnerHasher::hash_no_pad(&public_inputs)" (upstream prover.rs); of the empty list
  // (the fib-64 circuit: src/p3/mod.rs:264 prints []) it is four zeros
  const RHash pi_hash = ref_hash_no_pad(wr.public_inputs.data(), wr.public_inputs.size());
  const u64* pih = pi_hash.e;

  RPolyBatch wires = ref_commit_values(wr.wires, c.rate_bits, c.cap_height);
  T.wires_commit = now_s() - t0; t0 = now_s();

  RChallenger ch;
  ch.observe_hash(pre.circuit_digest);
  ch.observe_hash(pi_hash);
  ch.observe_cap(wires.tree.cap());
  std::vector<u64> betas(NC), gammas(NC);
  for (auto& b : betas) b = ch.challenge();
  for (auto& g : gammas) g = ch.challenge();

  std::vector<std::vector<u64>> zs_pp = ref_partial_products(c, wr.wires, betas, gammas);
  T.zs = now_s() - t0; t0 = now_s();
  RPolyBatch zs_batch = ref_commit_values(zs_pp, c.rate_bits, c.cap_height);
  T.zs_commit = now_s() - t0; t0 = now_s();
  ch.observe_cap(zs_batch.tree.cap());
  std::vector<u64> alphas(NC);
  for (auto& a : alphas) a = ch.challenge();

  std::vector<std::vector<u64>> qchunks = ref_quotient_chunks(c, pre.constants_sigmas, wires, zs_batch, betas, gammas, alphas, pih);
  T.quotient = now_s() - t0; t0 = now_s();
  RPolyBatch quot = ref_commit_coeffs(qchunks, c.rate_bits, c.cap_height);
  T.quotient_commit = now_s() - t0; t0 = now_s();
  ch.observe_cap(quot.tree.cap());
  RE2 zeta = ch.ext_challenge();
  if (re_eq(re_exp_pow2(zeta, c.degree_bits), re(1))) {
    if (msg) *msg = "Opening point is in the subgroup.";
    return 6;
  }
  const u64 g = rf_root_of_unity(c.degree_bits);
  RE2 zeta_next = re_muls(zeta, g);

  // openings (App. A.7)
  const RPolyBatch* oracles[4] = {&pre.constants_sigmas, &wires, &zs_batch, &quot};
  std::vector<std::vector<RE2>> ev(4);
  for (int o = 0; o < 4; o++) {
    ev[o].resize(oracles[o]->n_polys);
    parallel_for(oracles[o]->n_polys, [&](size_t b, size_t e) {
      for (size_t p = b; p < e; p++) ev[o][p] = eval_poly_ext(oracles[o]->coeffs[p], zeta);
    });
  }
  out = RProof();
  out.public_inputs = wr.public_inputs;
  out.wires_cap = wires.tree.cap();
  out.zs_cap = zs_batch.tree.cap();
  out.quotient_cap = quot.tree.cap();
  out.constants.assign(ev[0].begin(), ev[0].begin() + n_consts);
  out.sigmas.assign(ev[0].begin() + n_consts, ev[0].end());
  out.wires = ev[1];
  out.zs.assign(ev[2].begin(), ev[2].begin() + NC);
  out.pps.assign(ev[2].begin() + NC, ev[2].end());
  out.quotient = ev[3];
  for (int k = 0; k < NC; k++) out.zs_next.push_back(eval_poly_ext(zs_batch.coeffs[k], zeta_next));
  // observe openings: batch zeta = constants|sigmas|wires|zs|pps|quotient, batch g*zeta = zs_next
  for (auto* v : {&out.constants, &out.sigmas, &out.wires, &out.zs, &out.pps, &out.quotient, &out.zs_next})
    for (auto& e : *v) ch.observe_ext(e);
  T.openings = now_s() - t0; t0 = now_s();

  // FRI batching (App. A.7): final = X * (alpha^|B1| * q_B0 + q_B1)
  RE2 fri_alpha = ch.ext_challenge();
  std::vector<RE2> final_poly(n, re(0));
  {
    std::vector<const std::vector<u64>*> b0;
    for (int o = 0; o < 4; o++)
      for (auto& p : oracles[o]->coeffs) b0.push_back(&p);
    std::vector<const std::vector<u64>*> b1;
    for (int k = 0; k < NC; k++) b1.push_back(&zs_batch.coeffs[k]);
    struct B {
      std::vector<const std::vector<u64>*>* polys;
      RE2 point;
    } batches[2] = {{&b0, zeta}, {&b1, zeta_next}};
    std::vector<RE2> acc;  // running final poly (n-1 coefficients)
    for (auto& bt : batches) {
      std::vector<RE2> comp(n, re(0));
      std::vector<RE2> apow(bt.polys->size());
      RE2 ap = re(1);
      for (auto& a : apow) {
        a = ap;
        ap = re_mul(ap, fri_alpha);
      }
      parallel_for(n, [&](size_t b, size_t e) {
        for (size_t k = b; k < e; k++) {
          RE2 s = re(0);
          for (size_t j = 0; j < bt.polys->size(); j++) s = re_add(s, re_muls(apow[j], (*(*bt.polys)[j])[k]));
          comp[k] = s;
        }
      });
      // divide_by_linear: q_k = sum_{j>k} c_j z^(j-k-1)
      std::vector<RE2> q(n - 1);
      RE2 run = re(0);
      for (size_t k = n; k-- > 1;) {
        run = re_add(re_mul(run, bt.point), comp[k]);
        q[k - 1] = run;
      }
      if (acc.empty()) {
        acc = q;
      } else {
        RE2 shift = ap;  // alpha^(number of polys in this batch)
        for (size_t k = 0; k < acc.size(); k++) acc[k] = re_add(re_mul(acc[k], shift), q[k]);
      }
    }
    for (size_t k = 0; k + 1 < n; k++) final_poly[k + 1] = acc[k];  // multiply by X
  }
  {
    RFriParams fp{c.degree_bits, c.rate_bits, c.cap_height, c.arity_bits, c.pow_bits, c.num_queries};
    int st = ref_fri_prove(fp, final_poly, ch, oracles, out, nullptr, msg);
    if (st) return st;
  }
  T.fri = now_s() - t0;
  T.total = now_s() - t_start;
  if (tm) *tm = T;
  return 0;
}

// ---------------------------------------------------------------- verifier (App. A.11)
int ref_verify(const RCircuit& c, const RHash& circuit_digest, const std::vector<RHash>& cs_cap, const RProof& p,
               std::string* msg) {
  auto fail = [&](int code, const char* m) {
    if (msg) *msg = m;
    return code;
  };
  const size_t n = c.n();
  const int NC = c.num_challenges, NP = c.num_partial_products, Q = c.quotient_degree_factor;
  const int lde_bits = c.degree_bits + c.rate_bits;
  const size_t big = (size_t)1 << lde_bits;
  if (p.public_inputs.size() != c.public_inputs.size()) return fail(1, "wrong number of public inputs");
  const RHash pi_hash = ref_hash_no_pad(p.public_inputs.data(), p.public_inputs.size());
  const u64* pih = pi_hash.e;
  RChallenger ch;
  ch.observe_hash(circuit_digest);
  ch.observe_hash(pi_hash);
  ch.observe_cap(p.wires_cap);
  std::vector<u64> betas(NC), gammas(NC), alphas(NC);
  for (auto& b : betas) b = ch.challenge();
  for (auto& g : gammas) g = ch.challenge();
  ch.observe_cap(p.zs_cap);
  for (auto& a : alphas) a = ch.challenge();
  ch.observe_cap(p.quotient_cap);
  RE2 zeta = ch.ext_challenge();
  for (auto* v : {&p.constants, &p.sigmas, &p.wires, &p.zs, &p.pps, &p.quotient, &p.zs_next})
    for (auto& e : *v) ch.observe_ext(e);
  RE2 fri_alpha = ch.ext_challenge();
  std::vector<RE2> fri_betas;
  for (auto& fc : p.fri_caps) {
    ch.observe_cap(fc);
    fri_betas.push_back(ch.ext_challenge());
  }
  for (auto& e : p.final_poly) ch.observe_ext(e);
  ch.observe(p.pow_witness);
  u64 pow_response = ch.challenge();
  std::vector<size_t> indices(c.num_queries);
  for (auto& x : indices) x = (size_t)(ch.challenge() % big);

  // vanishing(zeta) == Z_H(zeta) * sum_k zeta^(n k) t_k(zeta)
  {
    std::vector<FE> consts, sig, wv, z, zn, pp;
    for (auto& e : p.constants) consts.push_back(FE{e});
    for (auto& e : p.sigmas) sig.push_back(FE{e});
    for (auto& e : p.wires) wv.push_back(FE{e});
    for (auto& e : p.zs) z.push_back(FE{e});
    for (auto& e : p.zs_next) zn.push_back(FE{e});
    for (auto& e : p.pps) pp.push_back(FE{e});
    RE2 zeta_n = re_exp_pow2(zeta, c.degree_bits);
    RE2 z_h = re_sub(zeta_n, re(1));
    RE2 l0 = re_mul(z_h, re_inv(re_muls(re_sub(zeta, re(1)), (u64)n)));
    FE van[8];
    ref_eval_vanishing<FE>(c, FE{zeta}, FE{l0}, consts.data(), sig.data(), wv.data(), z.data(), zn.data(), pp.data(),
                           betas.data(), gammas.data(), alphas.data(), pih, van);
    for (int i = 0; i < NC; i++) {
      RE2 t = re(0);
      for (int k = Q; k-- > 0;) t = re_add(re_mul(t, zeta_n), p.quotient[i * Q + k]);
      if (!re_eq(van[i].v, re_mul(z_h, t))) return fail(10, "vanishing polynomial identity fails at zeta");
    }
  }
  // FRI
  if (leading_zeros64(pow_response) < c.pow_bits) return fail(11, "invalid proof of work");
  if ((int)p.queries.size() != c.num_queries) return fail(12, "wrong number of query rounds");
  const u64 g = rf_root_of_unity(c.degree_bits);
  RE2 zeta_next = re_muls(zeta, g);
  // precomputed reduced openings
  std::vector<RE2> batch0;
  for (auto* v : {&p.constants, &p.sigmas, &p.wires, &p.zs, &p.pps, &p.quotient})
    batch0.insert(batch0.end(), v->begin(), v->end());
  auto reduce = [&](const std::vector<RE2>& v) {
    RE2 acc = re(0);
    for (size_t i = v.size(); i-- > 0;) acc = re_add(re_mul(acc, fri_alpha), v[i]);
    return acc;
  };
  RE2 red0 = reduce(batch0), red1 = reduce(p.zs_next);
  const std::vector<RHash>* caps[4] = {&cs_cap, &p.wires_cap, &p.zs_cap, &p.quotient_cap};
  for (int qi = 0; qi < c.num_queries; qi++) {
    const RProof::Query& q = p.queries[qi];
    size_t x_index = indices[qi];
    for (int o = 0; o < 4; o++)
      if (!ref_merkle_verify(q.initial_leaf[o], x_index, *caps[o], q.initial_path[o]))
        return fail(13, "initial Merkle proof fails");
    u64 sx = rf_mul(7, rf_pow(rf_root_of_unity(lde_bits), rbits(x_index, lde_bits)));
    // fri_combine_initial
    std::vector<RE2> e0;
    for (int o = 0; o < 4; o++)
      for (u64 v : q.initial_leaf[o]) e0.push_back(re(v));
    std::vector<RE2> e1;
    for (int k = 0; k < NC; k++) e1.push_back(re(q.initial_leaf[2][k]));
    RE2 sum = re_mul(re_sub(reduce(e0), red0), re_inv(re_sub(re(sx), zeta)));
    RE2 apow = re_pow(fri_alpha, e1.size());
    sum = re_add(re_mul(sum, apow), re_mul(re_sub(reduce(e1), red1), re_inv(re_sub(re(sx), zeta_next))));
    RE2 old_eval = re_muls(sum, sx);  // multiply by X (upstream PR #436)
    u64 subgroup_x = sx;
    for (size_t l = 0; l < c.arity_bits.size(); l++) {
      const int ab = c.arity_bits[l];
      const size_t arity = (size_t)1 << ab;
      const std::vector<RE2>& evals = q.step_evals[l];
      size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
      if (!re_eq(evals[within], old_eval)) return fail(14, "FRI layer evaluation inconsistent");
      // compute_evaluation: interpolate {(coset_start*g^i, evals_rev[i])} at beta
      u64 ga = rf_root_of_unity(ab);
      std::vector<RE2> ev(arity);
      for (size_t i = 0; i < arity; i++) ev[rbits(i, ab)] = evals[i];
      size_t rev_within = rbits(within, ab);
      u64 coset_start = rf_mul(subgroup_x, rf_pow(ga, arity - rev_within));
      std::vector<u64> xs(arity);
      u64 y = 1;
      for (size_t i = 0; i < arity; i++) {
        xs[i] = rf_mul(coset_start, y);
        y = rf_mul(y, ga);
      }
      RE2 acc = re(0);
      for (size_t i = 0; i < arity; i++) {
        RE2 numer = re(1);
        u64 denom = 1;
        for (size_t j = 0; j < arity; j++)
          if (j != i) {
            numer = re_mul(numer, re_sub(fri_betas[l], re(xs[j])));
            denom = rf_mul(denom, rf_sub(xs[i], xs[j]));
          }
        acc = re_add(acc, re_mul(ev[i], re_muls(numer, rf_inv(denom))));
      }
      old_eval = acc;
      if (!ref_merkle_verify(flatten_exts(evals), coset_index, p.fri_caps[l], q.step_path[l]))
        return fail(15, "FRI layer Merkle proof fails");
      for (int k = 0; k < ab; k++) subgroup_x = rf_mul(subgroup_x, subgroup_x);
      x_index = coset_index;
    }
    if (!re_eq(eval_ext_poly(p.final_poly, re(subgroup_x)), old_eval)) return fail(16, "final polynomial mismatch");
  }
  (void)n;
  (void)NP;
  return 0;
}
