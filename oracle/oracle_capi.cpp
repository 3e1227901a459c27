// ORACLE -- test infrastructure only.  C entry points (ctypes) over the CPU restatement, used by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the product.
#include <string.h>
#include "ref_fft.h"
#include "ref_hash.h"
#include "ref_gates.h"
#include "ref_prover.h"
#include "ref_hash_x8.h"
#include <atomic>
#include <chrono>
#include <memory>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <thread>

extern "C" {

void p25o_poseidon_permute(u64* states, size_t n) {
  for (size_t i = 0; i < n; i++) ref_poseidon(states + 12 * i);
}
void p25o_poseidon_permute_naive(u64* states, size_t n) {
  for (size_t i = 0; i < n; i++) ref_poseidon_naive(states + 12 * i);
}
void p25o_poseidon2_permute(u64* states, size_t n) {
  for (size_t i = 0; i < n; i++) ref_poseidon2(states + 12 * i);
}
void p25o_poseidon2_trace(u64* state, u64* trace) { ref_poseidon2_trace(state, trace); }
void p25o_poseidon_trace(u64* state, u64* trace) { ref_poseidon_trace(state, trace); }
void p25o_poseidon_fast_partial_inputs(u64* state, u64* partial_in) { ref_poseidon_fast_partial_inputs(state, partial_in); }
void p25o_hash_no_pad(const u64* in, size_t n, u64* out4) {
  RHash h = ref_hash_no_pad(in, n);
  memcpy(out4, h.e, 32);
}
u64 p25o_mul(u64 a, u64 b) { return rf_mul(a, b); }
u64 p25o_inv(u64 a) { return rf_inv(a); }

// leaves row-major [n_leaves][width] (upstream orientation).  tree_out nullable:
// all levels, leaf digests first, 4 words per node.
void p25o_merkle_commit(const u64* leaves_rm, size_t n_leaves, size_t width, unsigned cap_height,
                        u64* cap_out, u64* tree_out) {
  std::vector<std::vector<u64>> leaves(n_leaves);
  for (size_t i = 0; i < n_leaves; i++) leaves[i].assign(leaves_rm + i * width, leaves_rm + (i + 1) * width);
  RMerkleTree t = ref_merkle_build(leaves, cap_height);
  memcpy(cap_out, t.cap().data(), t.cap().size() * 32);
  if (tree_out)
    for (auto& lv : t.levels) {
      memcpy(tree_out, lv.data(), lv.size() * 32);
      tree_out += lv.size() * 4;
    }
}

// PolynomialBatch::from_values / from_coeffs.  polys[n_polys][n].
// coeffs_out[n_polys][n]; lde_out[n_polys][n << rate_bits] in BIT-REVERSED index order (the order
// upstream's leaves are in after transpose + reverse_index_bits); cap_out[2^cap][4].
void p25o_lde_commit(const u64* polys, unsigned log_n, size_t n_polys, int from_coeffs,
                     unsigned rate_bits, unsigned cap_height, u64* coeffs_out, u64* lde_out,
                     u64* cap_out) {
  const size_t n = (size_t)1 << log_n, big = n << rate_bits;
  std::vector<std::vector<u64>> leaves(big, std::vector<u64>(n_polys));
  for (size_t p = 0; p < n_polys; p++) {
    std::vector<u64> c(polys + p * n, polys + (p + 1) * n);
    if (!from_coeffs) ref_ifft(c);
    if (coeffs_out) memcpy(coeffs_out + p * n, c.data(), n * 8);
    std::vector<u64> v = ref_lde_values(c, rate_bits, 7);
    for (size_t i = 0; i < big; i++) {
      size_t r = rbits(i, log_n + rate_bits);
      leaves[r][p] = v[i];
      if (lde_out) lde_out[p * big + r] = v[i];
    }
  }
  if (cap_out) {
    RMerkleTree t = ref_merkle_build(leaves, cap_height);
    memcpy(cap_out, t.cap().data(), t.cap().size() * 32);
  }
}


// ---------------------------------------------------------------- circuit-level entry points
struct OracleCircuit {
  RCircuit c;
  std::unique_ptr<RPrecomputed> pre;
};
static void put_msg(char* dst, size_t cap, const std::string& m) {
  if (dst && cap) snprintf(dst, cap, "%s", m.c_str());
}

void p25o_set_threads(int n) { ref_set_threads(n); }
// tuned cpu_baseline leg: AVX-512 Merkle hashing (ref_hash_x8.cpp); returns 1 if in effect (0: asked off, or no AVX-512)
int p25o_set_tuned(int on) { return ref_set_tuned(on); }
int p25o_x8_available() { return ref_x8_available() ? 1 : 0; }
void p25o_poseidon_permute_x8(u64* states) { ref_poseidon_x8((u64(*)[12])states); }

void* p25o_circuit_load(const unsigned char* blob, size_t len) {
  try {
    auto* oc = new OracleCircuit();
    oc->c = ref_circuit_parse(blob, len);
    return oc;
  } catch (...) {
    return nullptr;
  }
}
void p25o_circuit_free(void* h) { delete (OracleCircuit*)h; }
int p25o_circuit_info(void* h, u64* out /*[8]*/) {
  auto* oc = (OracleCircuit*)h;
  out[0] = oc->c.degree_bits; out[1] = oc->c.num_wires; out[2] = oc->c.num_inputs; out[3] = oc->c.gens.size();
  out[4] = ref_proof_words(oc->c); out[5] = oc->c.num_cs(); out[6] = oc->c.gates.size(); out[7] = oc->c.num_gate_constraints;
  return 0;
}
// wires_out: [num_wires][n]
int p25o_witness(void* h, const u64* inputs, u64 seed, u64* wires_out, char* msg, size_t msglen) {
  auto* oc = (OracleCircuit*)h;
  RWitnessResult r = ref_generate_witness(oc->c, inputs, seed);
  put_msg(msg, msglen, r.message);
  if (r.status) return r.status;
  const size_t n = oc->c.n();
  for (int col = 0; col < oc->c.num_wires; col++) memcpy(wires_out + (size_t)col * n, r.wires[col].data(), n * 8);
  return 0;
}
// Witness generation that does not stop at a copy-constraint conflict: the conflicting partition keeps its first
// value, the wires are returned, the status is still 4 (see ref_generate_witness keep_going).
int p25o_witness_forced(void* h, const u64* inputs, u64 seed, u64* wires_out, char* msg, size_t msglen) {
  auto* oc = (OracleCircuit*)h;
  RWitnessResult r = ref_generate_witness(oc->c, inputs, seed, nullptr, true);
  put_msg(msg, msglen, r.message);
  if (r.wires.empty()) return r.status ? r.status : 5;
  const size_t n = oc->c.n();
  for (int col = 0; col < oc->c.num_wires; col++) memcpy(wires_out + (size_t)col * n, r.wires[col].data(), n * 8);
  return r.status;
}
// Evaluates every row's own gate on the witness; returns the number of non-zero constraints
// (0 = witness satisfies the circuit) and reports the first offender.
long p25o_check_constraints(void* h, const u64* wires, char* msg, size_t msglen) {
  auto* oc = (OracleCircuit*)h;
  const RCircuit& c = oc->c;
  const size_t n = c.n();
  long bad = 0;
  std::vector<FB> w(c.num_wires), out(c.num_gate_constraints + 8);
  // public-inputs hash: what the PublicInputGate row of this witness holds (that it IS the hash of the registered
  // inputs is enforced by the hashing rows behind it and checked by the prover / verifier pair)
  FB pih[4] = {FB{0}, FB{0}, FB{0}, FB{0}};
  if (c.pi_row >= 0)
    for (int i = 0; i < 4; i++) pih[i] = FB{wires[(size_t)i * n + (size_t)c.pi_row]};
  for (size_t row = 0; row < n; row++) {
    for (int col = 0; col < c.num_wires; col++) w[col] = FB{wires[(size_t)col * n + row]};
    FB k[2] = {FB{c.constants_sigmas[c.num_selectors][row]}, FB{c.constants_sigmas[c.num_selectors + 1][row]}};
    int nc = ref_eval_gate<FB>(c.row_kind[row], w.data(), k, pih, out.data());
    for (int j = 0; j < nc; j++)
      if (out[j].v != 0) {
        if (!bad) put_msg(msg, msglen, "row " + std::to_string(row) + " gate kind " + std::to_string(c.row_kind[row]) +
                                           " constraint " + std::to_string(j) + " = " + std::to_string(out[j].v));
        bad++;
      }
  }
  // copy constraints: every routed wire equals its representative's value
  std::vector<u64> rep_val(c.num_targets(), 0);
  std::vector<unsigned char> seen(c.num_targets(), 0);
  for (size_t row = 0; row < n; row++)
    for (int col = 0; col < c.num_routed; col++) {
      u32 r = c.rep[row * c.num_wires + col];
      u64 v = wires[(size_t)col * n + row];
      if (!seen[r]) {
        seen[r] = 1;
        rep_val[r] = v;
      } else if (rep_val[r] != v) {
        if (!bad) put_msg(msg, msglen, "copy constraint violated at row " + std::to_string(row) + " col " + std::to_string(col));
        bad++;
      }
    }
  return bad;
}
int p25o_precompute(void* h) {
  auto* oc = (OracleCircuit*)h;
  if (!oc->pre) oc->pre.reset(new RPrecomputed(ref_precompute(oc->c)));
  return 0;
}
void p25o_circuit_digest(void* h, u64* digest4, u64* cs_cap /*[2^cap][4]*/) {
  auto* oc = (OracleCircuit*)h;
  p25o_precompute(h);
  memcpy(digest4, oc->pre->circuit_digest.e, 32);
  if (cs_cap) memcpy(cs_cap, oc->pre->constants_sigmas.tree.cap().data(), oc->pre->constants_sigmas.tree.cap().size() * 32);
}
size_t p25o_proof_words(void* h) { return ref_proof_words(((OracleCircuit*)h)->c); }
// timings_out[9]: witness, wires_commit, zs, zs_commit, quotient, quotient_commit, openings, fri, total (seconds)
int p25o_prove(void* h, const u64* inputs, u64 seed, u64* proof_out, double* timings_out, char* msg, size_t msglen) {
  auto* oc = (OracleCircuit*)h;
  p25o_precompute(h);
  RProof pr;
  RTimings tm;
  std::string m;
  int st = ref_prove(oc->c, *oc->pre, inputs, seed, pr, &tm, &m);
  put_msg(msg, msglen, m);
  if (st) return st;
  std::vector<u64> flat = ref_proof_flatten(oc->c, pr);
  if (flat.size() != ref_proof_words(oc->c)) {
    put_msg(msg, msglen, "internal: proof size mismatch");
    return 7;
  }
  memcpy(proof_out, flat.data(), flat.size() * 8);
  if (timings_out) {
    double t[9] = {tm.witness, tm.wires_commit, tm.zs, tm.zs_commit, tm.quotient, tm.quotient_commit, tm.openings, tm.fri, tm.total};
    memcpy(timings_out, t, sizeof(t));
  }
  return 0;
}
// One row's constraints over F_p^2 (upstream Gate::eval_unfiltered; the verifier's evaluator): wires[num_wires][2],
// consts[2][2], pih[4] -> out[n][2]; returns n.  With base = 1 the wires' second components are ignored and the
// base-field evaluator runs (out[n][2] with zero second components): the two must agree on base inputs.
int p25o_eval_gate(unsigned kind, int num_wires, const u64* wires, const u64* consts, const u64* pih, int base, u64* out) {
  if (base) {
    std::vector<FB> w(num_wires), o(512);
    for (int i = 0; i < num_wires; i++) w[i] = FB{wires[2 * i]};
    FB k[2] = {FB{consts[0]}, FB{consts[2]}}, ph[4] = {FB{pih[0]}, FB{pih[1]}, FB{pih[2]}, FB{pih[3]}};
    int n = ref_eval_gate<FB>(kind, w.data(), k, ph, o.data());
    for (int j = 0; j < n; j++) {
      out[2 * j] = o[j].v;
      out[2 * j + 1] = 0;
    }
    return n;
  }
  std::vector<FE> w(num_wires), o(512);
  for (int i = 0; i < num_wires; i++) w[i] = FE{RE2{wires[2 * i], wires[2 * i + 1]}};
  FE k[2] = {FE{RE2{consts[0], consts[1]}}, FE{RE2{consts[2], consts[3]}}};
  FE ph[4] = {FE::from(pih[0]), FE::from(pih[1]), FE::from(pih[2]), FE::from(pih[3])};
  int n = ref_eval_gate<FE>(kind, w.data(), k, ph, o.data());
  for (int j = 0; j < n; j++) {
    out[2 * j] = o[j].v.a;
    out[2 * j + 1] = o[j].v.b;
  }
  return n;
}

// ---------------------------------------------------------------- isolated stages (parity tests of a6-a10)
// Challenger script: for each segment observe seg_len[k] words of `obs` (consumed in order), then draw
// n_chal[k] challenges into `out` (appended in order).  upstream iop/challenger.rs.
void p25o_transcript(const u64* obs, const u32* seg_len, const u32* n_chal, size_t n_seg, u64* out) {
  RChallenger ch;
  for (size_t k = 0; k < n_seg; k++) {
    for (u32 i = 0; i < seg_len[k]; i++) ch.observe(*obs++);
    for (u32 i = 0; i < n_chal[k]; i++) *out++ = ch.challenge();
  }
}
// rows of the Z/partial-products matrix and of the quotient-chunk matrix
void p25o_stage_shapes(void* h, u64* nz, u64* nq) {
  const RCircuit& c = ((OracleCircuit*)h)->c;
  if (nz) *nz = (u64)c.num_challenges * (1 + c.num_partial_products);
  if (nq) *nq = (u64)c.num_challenges * c.quotient_degree_factor;
}
// wires[num_wires][n] -> out[NC * (1 + NP)][n]
void p25o_partial_products(void* h, const u64* wires, const u64* betas, const u64* gammas, u64* out) {
  auto* oc = (OracleCircuit*)h;
  const RCircuit& c = oc->c;
  const size_t n = c.n();
  std::vector<std::vector<u64>> w(c.num_wires);
  for (int col = 0; col < c.num_wires; col++) w[col].assign(wires + (size_t)col * n, wires + (size_t)(col + 1) * n);
  std::vector<u64> b(betas, betas + c.num_challenges), g(gammas, gammas + c.num_challenges);
  auto z = ref_partial_products(c, w, b, g);
  for (size_t k = 0; k < z.size(); k++) memcpy(out + k * n, z[k].data(), n * 8);
}
// wires[num_wires][n], zs_pp[NC*(1+NP)][n] (values) -> out[NC * Q][n] quotient chunk coefficients
void p25o_quotient(void* h, const u64* wires, const u64* zs_pp, const u64* betas, const u64* gammas, const u64* alphas,
                   u64* out) {
  auto* oc = (OracleCircuit*)h;
  p25o_precompute(h);
  const RCircuit& c = oc->c;
  const size_t n = c.n();
  const int nz = c.num_challenges * (1 + c.num_partial_products);
  std::vector<std::vector<u64>> w(c.num_wires), z(nz);
  for (int col = 0; col < c.num_wires; col++) w[col].assign(wires + (size_t)col * n, wires + (size_t)(col + 1) * n);
  for (int k = 0; k < nz; k++) z[k].assign(zs_pp + (size_t)k * n, zs_pp + (size_t)(k + 1) * n);
  RPolyBatch wb = ref_commit_values(w, c.rate_bits, c.cap_height), zb = ref_commit_values(z, c.rate_bits, c.cap_height);
  std::vector<u64> b(betas, betas + c.num_challenges), g(gammas, gammas + c.num_challenges),
      a(alphas, alphas + c.num_challenges);
  // the public-inputs hash the PublicInputGate compares with: what that row of the given witness holds
  u64 pih[4] = {0, 0, 0, 0};
  if (c.pi_row >= 0)
    for (int i = 0; i < 4; i++) pih[i] = wires[(size_t)i * n + (size_t)c.pi_row];
  auto q = ref_quotient_chunks(c, oc->pre->constants_sigmas, wb, zb, b, g, a, pih);
  for (size_t k = 0; k < q.size(); k++) memcpy(out + k * n, q[k].data(), n * 8);
}
// coeffs[n_polys][n] at the extension point point*scale (Horner) -> out[n_polys][2]   (upstream PolynomialCoeffs::eval)
void p25o_eval_polys(const u64* coeffs, size_t n_polys, size_t n, const u64* point, u64 scale, u64* out) {
  RE2 z = re_muls(RE2{point[0], point[1]}, scale);
  for (size_t p = 0; p < n_polys; p++) {
    RE2 acc = re(0);
    for (size_t k = n; k-- > 0;) acc = re_add(re_mul(acc, z), re(coeffs[p * n + k]));
    out[2 * p] = acc.a;
    out[2 * p + 1] = acc.b;
  }
}
// FRI on one batched polynomial: coeffs[2][2^log_n] (extension components), transcript initialised by observing
// `seed`.  out: CAP[n_layers] | betas E[n_layers] | final_poly E[..] | pow_witness | indices u64[num_queries] |
// per query, per layer {evals E[2^arity], siblings H[..]}.  Returns words written (0 on failure).
size_t p25o_fri_prove(const u64* coeffs, unsigned log_n, unsigned rate_bits, unsigned cap_height, const int* arity_bits,
                      size_t n_layers, unsigned pow_bits, unsigned num_queries, const u64* seed, size_t n_seed, u64* out,
                      size_t cap) {
  const size_t n = (size_t)1 << log_n;
  RFriParams fp{(int)log_n, (int)rate_bits, (int)cap_height, std::vector<int>(arity_bits, arity_bits + n_layers),
                (int)pow_bits, (int)num_queries};
  std::vector<RE2> poly(n);
  for (size_t i = 0; i < n; i++) poly[i] = RE2{coeffs[i], coeffs[n + i]};
  RChallenger ch;
  for (size_t i = 0; i < n_seed; i++) ch.observe(seed[i]);
  // the betas are not part of a proof: recover them by replaying the transcript on a copy
  RChallenger replay = ch;
  RProof pr;
  std::vector<size_t> idx;
  std::string m;
  if (ref_fri_prove(fp, poly, ch, nullptr, pr, &idx, &m)) return 0;
  std::vector<u64> o;
  for (auto& c : pr.fri_caps)
    for (auto& hsh : c) o.insert(o.end(), hsh.e, hsh.e + 4);
  for (auto& c : pr.fri_caps) {
    replay.observe_cap(c);
    RE2 b = replay.ext_challenge();
    o.push_back(b.a);
    o.push_back(b.b);
  }
  for (auto& e : pr.final_poly) {
    o.push_back(e.a);
    o.push_back(e.b);
  }
  o.push_back(pr.pow_witness);
  for (size_t x : idx) o.push_back((u64)x);
  for (auto& q : pr.queries)
    for (size_t l = 0; l < q.step_evals.size(); l++) {
      for (auto& e : q.step_evals[l]) {
        o.push_back(e.a);
        o.push_back(e.b);
      }
      for (auto& hsh : q.step_path[l]) o.insert(o.end(), hsh.e, hsh.e + 4);
    }
  if (o.size() > cap) return 0;
  memcpy(out, o.data(), o.size() * 8);
  return o.size();
}

// CPU-baseline leg "one proof per core": n_proofs independent proofs on n_threads host threads, each
// proof single-threaded (as the reference build: Cargo.toml:15-18 has no `parallel` feature) and each
// thread pinned to its own CPU of the process's affinity set.  inputs[n_proofs][num_inputs],
// proofs_out[n_proofs][proof_words] (nullable), statuses[n_proofs], per_proof_s[n_proofs] (nullable:
// wall seconds of each proof).  Returns wall seconds of the whole run.
double p25o_prove_many_grouped(void* h, const u64* inputs, const u64* seeds, size_t n_proofs, int n_threads,
                               int threads_per_proof, u64* proofs_out, int* statuses, double* per_proof_s);
double p25o_prove_many(void* h, const u64* inputs, const u64* seeds, size_t n_proofs, int n_threads,
                       u64* proofs_out, int* statuses, double* per_proof_s) {
  return p25o_prove_many_grouped(h, inputs, seeds, n_proofs, n_threads, 1, proofs_out, statuses, per_proof_s);
}
// Same with `threads_per_proof` host threads per proof (n_threads proofs in flight, each spreading its loops over
// its own group through the persistent pool): the whole-host configuration that is not limited by n_threads
// working sets of ~4 GB each competing for the memory system.  Proving threads are pinned only when
// threads_per_proof == 1.
double p25o_prove_many_grouped(void* h, const u64* inputs, const u64* seeds, size_t n_proofs, int n_threads,
                               int threads_per_proof, u64* proofs_out, int* statuses, double* per_proof_s) {
  auto* oc = (OracleCircuit*)h;
  p25o_precompute(h);
  const size_t ni = oc->c.num_inputs, pw = ref_proof_words(oc->c);
  if (n_threads < 1) n_threads = 1;
  std::vector<int> cpus;
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0)
      for (int i = 0; i < CPU_SETSIZE; i++)
        if (CPU_ISSET(i, &set)) cpus.push_back(i);
  }
  std::atomic<size_t> next{0};
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++)
    th.emplace_back([&, t] {
      if (!cpus.empty() && threads_per_proof <= 1) {
        cpu_set_t one;
        CPU_ZERO(&one);
        CPU_SET(cpus[t % cpus.size()], &one);
        (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
      }
      ref_set_thread_local_threads(threads_per_proof < 1 ? 1 : threads_per_proof);
      for (;;) {
        size_t i = next.fetch_add(1);
        if (i >= n_proofs) break;
        auto a = std::chrono::steady_clock::now();
        RProof pr;
        std::string m;
        int st = ref_prove(oc->c, *oc->pre, inputs + i * ni, seeds ? seeds[i] : (u64)i, pr, nullptr, &m);
        statuses[i] = st;
        if (!st && proofs_out) {
          std::vector<u64> flat = ref_proof_flatten(oc->c, pr);
          memcpy(proofs_out + i * pw, flat.data(), pw * 8);
        }
        if (per_proof_s) per_proof_s[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
      }
    });
  for (auto& x : th) x.join();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Same with explicit RandomValueGenerator values filler[num_random_fill] instead of the seed.
int p25o_prove_filler(void* h, const u64* inputs, const u64* filler, u64* proof_out, char* msg, size_t msglen) {
  auto* oc = (OracleCircuit*)h;
  p25o_precompute(h);
  RProof pr;
  std::string m;
  int st = ref_prove(oc->c, *oc->pre, inputs, 0, pr, nullptr, &m, filler);
  put_msg(msg, msglen, m);
  if (st) return st;
  std::vector<u64> flat = ref_proof_flatten(oc->c, pr);
  memcpy(proof_out, flat.data(), flat.size() * 8);
  return 0;
}
size_t p25o_num_random_fill(void* h) { return ((OracleCircuit*)h)->c.num_random_fill; }

// digest4 / cs_cap: the verifier-side circuit data (VerifierOnlyCircuitData); pass the oracle's own
// (p25o_circuit_digest) or the product's to cross-check.
int p25o_verify(void* h, const u64* digest4, const u64* cs_cap, const u64* proof_words, char* msg, size_t msglen) {
  auto* oc = (OracleCircuit*)h;
  RProof pr = ref_proof_unflatten(oc->c, proof_words);
  RHash d;
  memcpy(d.e, digest4, 32);
  std::vector<RHash> cap((size_t)1 << oc->c.cap_height);
  memcpy(cap.data(), cs_cap, cap.size() * 32);
  std::string m;
  int st = ref_verify(oc->c, d, cap, pr, &m);
  put_msg(msg, msglen, m);
  return st;
}

}  // extern "C"
