// See prover.h.
#include "prover.h"
#include <algorithm>
#include <map>
#include <mutex>
#include <string.h>
#include "poseidon.h"

namespace p25 {

// Proving streams come from ONE pool per device, shared by every circuit of the process.  HIP streams are multiplexed
// onto GPU_MAX_HW_QUEUES hardware queues (24, capi.hip), and two streams in one hardware queue run in order: with a
// stream set per circuit, a second circuit that is merely ALIVE (the leaf circuit's 9 idle streams beside an
// aggregation circuit's 17) cost the active one 9 % (120.7 against 131.8 aggregate proofs/s, round 4,
// profiles/r04_stream_pool.txt), and more hardware queues cost more than they return.  Context i of a circuit takes
// pool stream (first + i) mod P, `first` = where the pool's cursor stood when the circuit made its first context, so the
// few contexts of the upper levels of an aggregation tree land on different streams; P = STREAM_POOL (16), or the
// largest proofs-in-flight count any circuit has asked for.  Two circuits proving at the same time interleave their
// proofs on the shared streams (every ordering inside the library is by events, which stay correct -- conservatively
// so -- when a stream carries another circuit's work too).  Pool streams live as long as the process.
constexpr size_t STREAM_POOL = 16;
namespace {
struct StreamPool {
  std::mutex mu;
  std::map<int, std::vector<hipStream_t>> per_device;
  size_t cursor = 0;
  size_t width = STREAM_POOL;
  // a block of `count` consecutive pool positions (the caller's contexts first .. first + count - 1)
  size_t reserve(size_t count, size_t want_width) {
    std::lock_guard<std::mutex> l(mu);
    if (want_width > width) width = want_width;
    size_t first = cursor % width;
    cursor += count;
    return first;
  }
  // A circuit asked for more proofs in flight after its first context existed (p25_circuit_set_streams(32) on a live
  // circuit): the pool grows with it, so that contexts i and i + 16 do not silently share a stream.
  void widen(size_t want_width) {
    std::lock_guard<std::mutex> l(mu);
    if (want_width > width) width = want_width;
  }
  // Main streams (a circuit's witness passes): two per device.  The first circuit of the process to ask keeps one to
  // itself for as long as it lives (the leaf circuit of a batch prover, whose passes run under its own proving all the
  // time); every other circuit shares the second (the levels of an aggregation tree: a pass of at most 32 proofs per
  // step each).  The library then never holds more than 16 + 2 streams, whatever number of circuits is alive -- with a
  // main stream per circuit the pipelined tree lost 2.5 % as soon as the host added its gather stream and RCCL's
  // (profiles/r04_stream_pool.txt).  FIFO order on a shared stream is the order of the calls, which is the order of
  // their dependencies, and every wait is for an event recorded earlier: sharing adds no wait that could not end.
  struct Mains {
    hipStream_t st[2] = {nullptr, nullptr};
    size_t refs[2] = {0, 0};
  };
  std::map<int, Mains> mains;
  hipStream_t main_acquire(MainStreamLease& lease) {
    std::lock_guard<std::mutex> l(mu);
    int dev = 0;
    P25_HIP(hipGetDevice(&dev));
    Mains& m = mains[dev];
    const int which = m.refs[0] == 0 ? 0 : 1;
    if (!m.st[which]) P25_HIP(hipStreamCreate(&m.st[which]));
    m.refs[which]++;
    lease.slot = which;      // only now: a lease that failed to form gives nothing back
    lease.device = dev;
    return m.st[which];
  }
  void main_release(const MainStreamLease& lease) {
    std::lock_guard<std::mutex> l(mu);
    auto it = mains.find(lease.device);
    if (it != mains.end() && lease.slot >= 0 && lease.slot < 2 && it->second.refs[lease.slot]) it->second.refs[lease.slot]--;
  }
  hipStream_t at(size_t pos) {
    std::lock_guard<std::mutex> l(mu);
    int dev = 0;
    P25_HIP(hipGetDevice(&dev));
    auto& v = per_device[dev];
    const size_t k = pos % width;
    if (v.size() <= k) v.resize(k + 1, nullptr);
    if (!v[k]) P25_HIP(hipStreamCreate(&v[k]));
    return v[k];
  }
};
// never destroyed: circuits of static storage duration in a host may outlive any static of this library
StreamPool& g_stream_pool = *new StreamPool;
}  // namespace
MainStreamLease::~MainStreamLease() {
  if (slot >= 0) g_stream_pool.main_release(*this);
}

DevMem::DevMem(size_t w) : words(w) {
  if (w) P25_HIP(hipMalloc(&p, w * sizeof(u64)));
}
DevMem::~DevMem() {
  if (p) (void)hipFree(p);
}
DevMem& DevMem::operator=(DevMem&& o) noexcept {
  if (this != &o) {
    if (p) (void)hipFree(p);
    p = o.p;
    words = o.words;
    o.p = nullptr;
  }
  return *this;
}

static uint32_t final_poly_len(const Circuit& c) {
  int db = c.degree_bits;
  for (int a : c.fri_reduction_arity_bits) db -= a;
  return 1u << db;
}

ProofLayout make_proof_layout(const Circuit& c) {
  ProofLayout L;
  const size_t capw = (size_t)4 << c.cfg.cap_height;
  const int NC = c.cfg.num_challenges, NP = c.num_partial_products;
  L.oracle_width[0] = (uint32_t)c.constants_sigmas.size();
  L.oracle_width[1] = (uint32_t)c.cfg.num_wires;
  L.oracle_width[2] = (uint32_t)(NC * (1 + NP));
  L.oracle_width[3] = (uint32_t)(NC * c.cfg.max_quotient_degree_factor);
  size_t o = 0;
  L.wires_cap = o; o += capw;
  L.zs_cap = o; o += capw;
  L.quotient_cap = o; o += capw;
  L.constants = o; o += 2 * (size_t)(c.num_selectors + c.cfg.num_constants);
  L.sigmas = o; o += 2 * (size_t)c.cfg.num_routed_wires;
  L.wires = o; o += 2 * (size_t)c.cfg.num_wires;
  L.zs = o; o += 2 * (size_t)NC;
  L.zs_next = o; o += 2 * (size_t)NC;
  L.pps = o; o += 2 * (size_t)NC * NP;
  L.quotient = o; o += 2 * (size_t)L.oracle_width[3];
  L.fri_caps = o; o += c.fri_reduction_arity_bits.size() * capw;
  L.queries = o;
  const int lde_bits = c.degree_bits + c.cfg.rate_bits;
  size_t per_q = 0;
  for (int k = 0; k < 4; k++) per_q += L.oracle_width[k] + 4 * (size_t)(lde_bits - c.cfg.cap_height);
  int bits = lde_bits;
  for (int a : c.fri_reduction_arity_bits) {
    bits -= a;
    per_q += 2 * ((size_t)1 << a) + 4 * (size_t)(bits - c.cfg.cap_height);
  }
  L.query_stride = per_q;
  o += per_q * c.cfg.num_query_rounds;
  L.final_poly_len = final_poly_len(c);
  L.final_poly = o; o += 2 * (size_t)L.final_poly_len;
  L.pow_witness = o; o += 1;
  L.num_public_inputs = (uint32_t)c.public_inputs.size();
  L.public_inputs = o; o += L.num_public_inputs;   // ProofWithPublicInputs::public_inputs, after the proof proper
  L.total = o;
  return L;
}

// ---------------------------------------------------------------- small glue kernels
__global__ void k_check_zeta(const u64* chal, uint32_t degree_bits, uint32_t* status) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  gl::E2 z{chal[CH_ZETA], chal[CH_ZETA + 1]};
  gl::E2 zn = gl::exp_pow2(z, degree_bits);
  if (zn.a == 1 && zn.b == 0) set_status(status, 6);  // "Opening point is in the subgroup."
}
__global__ void k_interleave(const u64* a, const u64* b, uint32_t m, u64* out) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) {
    out[2 * i] = a[i];
    out[2 * i + 1] = b[i];
  }
}
__global__ void k_finish(const u64* chal, int pow_bits, u64* proof_pow, uint32_t* status) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  u64 w = chal[CH_POW_WITNESS];
  *proof_pow = w;
  if (w == ~0ull || __clzll((long long)chal[CH_POW_RESPONSE]) < pow_bits) set_status(status, 7);
}

// ---------------------------------------------------------------- FRI (upstream fri/prover.rs `fri_proof`)
void FriWork::alloc(int log_n, int rate_bits, unsigned cap_height, const std::vector<int>& arity_bits) {
  size_t m = (size_t)1 << log_n;
  const size_t nl = arity_bits.size();
  if (nl > 8) throw std::invalid_argument("too many FRI layers");
  for (size_t l = 0; l <= nl; l++) {
    coeffs[l] = DevMem(2 * m);
    if (l < nl) {
      size_t nvals = m << rate_bits;
      vals[l] = DevMem(2 * nvals);
      tree[l] = DevMem(merkle_tree_words(nvals >> arity_bits[l], cap_height));
      m >>= arity_bits[l];
    }
  }
}

// Commit phase ("fold codewords in the commitment phase"), "find proof-of-work witness", query rounds.
// In: the batched polynomial's coefficients in w.coeffs[0] (components a | b, n each), the transcript, and the
// initial-oracle part of `qy` (n_oracles may be 0).  Out: caps, final polynomial, PoW witness and query openings
// in the flat proof at the offsets of `fo`; betas / PoW / query challenges in the challenge block.
void fri_commit_pow_query(NttTables& tables, FriWork& w, const FriShape& sh, Transcript* tr, u64* chal, QueryArgs qy,
                          u64* d_proof, const FriOffsets& fo, uint32_t* d_status, hipStream_t st, bool single_proof) {
  const size_t capw = (size_t)4 << sh.cap_height;
  auto d2d = [&](u64* dst, const u64* src, size_t words) {
    P25_HIP(hipMemcpyAsync(dst, src, words * 8, hipMemcpyDeviceToDevice, st));
  };
  size_t m = (size_t)1 << sh.log_n;
  int log_m = sh.log_n;
  u64 shift = gl::GENERATOR;
  const size_t nl = sh.arity_bits.size();
  for (size_t l = 0; l < nl; l++) {
    const int ab = sh.arity_bits[l];
    const size_t nvals = m << sh.rate_bits;
    // values of the current polynomial on shift*<w>, bit-reversed positions, components a | b
    ntt_lde_bitrev(tables, w.coeffs[l].p, m, w.vals[l].p, nvals, log_m, sh.rate_bits, 2, shift, st);
    const size_t n_leaves = nvals >> ab;
    launch_fri_leaf_hash(w.vals[l].p, w.vals[l].p + nvals, (uint32_t)n_leaves, ab, w.tree[l].p, st, single_proof);
    launch_tree_from_digests(w.tree[l].p, n_leaves, sh.cap_height, st, single_proof);
    const size_t ltw = merkle_tree_words(n_leaves, sh.cap_height);
    const u64* cap = w.tree[l].p + ltw - capw;
    d2d(d_proof + fo.caps + l * capw, cap, capw);
    launch_transcript(tr, 0, cap, (uint32_t)capw, chal + CH_FRI_BETAS + 2 * l, 2, st);
    launch_fri_fold(w.coeffs[l].p, w.coeffs[l].p + m, (uint32_t)(m >> ab), ab, chal + CH_FRI_BETAS + 2 * l,
                    w.coeffs[l + 1].p, w.coeffs[l + 1].p + (m >> ab), st);
    qy.arity_bits[l] = ab;
    qy.layer_va[l] = w.vals[l].p;
    qy.layer_vb[l] = w.vals[l].p + nvals;
    qy.layer_tree[l] = w.tree[l].p;
    shift = gl::pow(shift, (u64)1 << ab);
    m >>= ab;
    log_m -= ab;
  }
  // final polynomial (the coefficients that survive truncation by the rate)
  hipLaunchKernelGGL(k_interleave, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, w.coeffs[nl].p,
                     w.coeffs[nl].p + m, (uint32_t)m, d_proof + fo.final_poly);
  launch_transcript(tr, 0, d_proof + fo.final_poly, (uint32_t)(2 * m), chal, 0, st);
  qy.n_layers = (uint32_t)nl;
  // "find proof-of-work witness"
  launch_pow_search(tr, sh.pow_bits, chal + CH_POW_WITNESS, st);
  launch_transcript(tr, 0, chal + CH_POW_WITNESS, 1, chal + CH_POW_RESPONSE, 1 + sh.num_queries, st);
  hipLaunchKernelGGL(k_finish, dim3(1), dim3(1), 0, st, chal, sh.pow_bits, d_proof + fo.pow_witness, d_status);
  // query rounds
  qy.chal = chal;
  qy.num_queries = sh.num_queries;
  qy.lde_bits = sh.log_n + sh.rate_bits;
  qy.cap_height = sh.cap_height;
  qy.proof = d_proof;
  qy.query_offset = (uint32_t)fo.queries;
  qy.query_stride = (uint32_t)fo.query_stride;
  launch_queries(qy, st);
}

// ---------------------------------------------------------------- per-proof working set
struct DeviceCircuit::Ctx {
  DevMem wires_vals, wires_coeffs, tmp, wires_lde, wires_tree;
  DevMem zs_vals, zs_coeffs, zs_lde, zs_tree, zpp_chunk, zpp_tot, zpp_btot;
  DevMem q_vals, q_tmp, q_coeffs, q_lde, q_tree;
  DevMem transcript, chal, alpha_pows, eval_pows;
  DevMem preamble;  // circuit_digest[4] | public_inputs_hash[4] of the proof in flight: what the transcript absorbs first
  DevMem fri_comp, fri_scan;
  FriWork fri;
  DevMem proof, status;
  hipEvent_t ev[12];
  hipStream_t st = nullptr;   // a stream of the process-wide pool (never destroyed)
  // done[b]: recorded after the context's latest read of witness-value buffer b (DeviceCircuit::vals_[b])
  hipEvent_t done[2] = {nullptr, nullptr};
  hipEvent_t join = nullptr;   // DeviceCircuit::stream_join
  bool have_events = false;
  ~Ctx() {
    if (have_events)
      for (auto& e : ev) (void)hipEventDestroy(e);
    for (auto& e : done)
      if (e) (void)hipEventDestroy(e);
    if (join) (void)hipEventDestroy(join);
  }
};

DeviceCircuit::DeviceCircuit(Circuit c) : c_(std::move(c)) {
  if (c_.cfg.num_challenges != 2 || c_.cfg.rate_bits > 3 || c_.gates.size() > 16)
    throw std::invalid_argument("unsupported circuit configuration");
  if (c_.fri_reduction_arity_bits.size() > 8) throw std::invalid_argument("too many FRI layers");
  // register / LDS capacities of k_zpp_* and k_quotient (also enforced when a blob is imported)
  if (c_.num_partial_products + 1 > MAX_CHUNKS || c_.cfg.num_routed_wires > MAX_ROUTED ||
      c_.cfg.num_challenges * (2 + c_.num_partial_products) >= ALPHA_POWS || c_.num_gate_constraints > ALPHA_POWS)
    throw std::invalid_argument("circuit exceeds the permutation-argument / quotient kernels' capacities");
  stream_ = g_stream_pool.main_acquire(main_lease_);
  layout_ = make_proof_layout(c_);
  const size_t n = c_.degree();
  const int ncs = (int)c_.constants_sigmas.size();
  // witness program
  WitnessProgram wp = build_witness_program(c_);
  auto up32 = [&](const std::vector<uint32_t>& v) {
    DevMem m((v.size() + 1) / 2 + 1);
    P25_HIP(hipMemcpy(m.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    u64* p = m.p;
    owned_.push_back(std::move(m));
    return (uint32_t*)p;
  };
  {
    DevMem m(wp.gens.size() * sizeof(WitGen) / 8 + 1);
    P25_HIP(hipMemcpy(m.p, wp.gens.data(), wp.gens.size() * sizeof(WitGen), hipMemcpyHostToDevice));
    wp_.d_gens = (WitGen*)m.p;
    owned_.push_back(std::move(m));
  }
  wp_.d_args = up32(wp.args);
  wp_.d_input_slots = up32(wp.input_slots);
  wp_.d_input_first = up32(wp.input_first);
  wp_.d_wire_slot_cm = up32(wp.wire_slot_cm);
  wp_.level_start = wp.level_start;
  for (size_t l = 0; l + 1 < wp.level_start.size(); l++) {
    uint32_t b = wp.level_start[l], e = wp.level_start[l + 1];
    uint32_t kind = GEN_POSEIDON2, pb = e, pc = 0;
    for (int pass = 0; pass < 2 && !pc; pass++) {
      kind = pass == 0 ? GEN_POSEIDON2 : GEN_POSEIDON;
      for (uint32_t g = b; g < e; g++)
        if (wp.gens[g].kind == kind) {
          if (!pc) pb = g;
          pc++;
        }
    }
    wp_.level_p2_begin.push_back(pb);
    wp_.level_p2_count.push_back(pc);
    wp_.level_coop_kind.push_back(kind);
    // the cooperative kernel addresses the arguments of permutation k of the level at base + k * 135: check the layout
    for (uint32_t k = 0; k < pc; k++) {
      const WitGen& g = wp.gens[pb + k];
      if (g.kind != kind || g.n_deps != 13 || g.n_outs != 4 + 106 + 12 || g.arg_off != wp.gens[pb].arg_off + k * 135u)
        throw std::logic_error("witness program: permutation generators of a level are not laid out contiguously");
    }
    wp_.level_perm_arg_base.push_back(pc ? wp.gens[pb].arg_off : 0);
  }
  wp_.d_pi_slots = wp.pi_slots.empty() ? nullptr : up32(wp.pi_slots);
  wp_.n_public_inputs = (uint32_t)wp.pi_slots.size();
  wp_.n_inputs = (uint32_t)wp.input_slots.size();
  wp_.num_slots = wp.num_slots;
  wp_.num_random_fill = wp.num_random_fill;
  wp_.n_wire_elems = wp.wire_slot_cm.size();

  // constants_sigmas commitment (PolynomialBatch::from_values) -- once per circuit
  cs_vals_ = DevMem((size_t)ncs * n);
  for (int p = 0; p < ncs; p++)
    P25_HIP(hipMemcpy(cs_vals_.p + (size_t)p * n, c_.constants_sigmas[p].data(), n * 8, hipMemcpyHostToDevice));
  cs_coeffs_ = DevMem((size_t)ncs * n);
  cs_lde_ = DevMem((size_t)ncs * big());
  const size_t tw = merkle_tree_words(big(), c_.cfg.cap_height);
  cs_tree_ = DevMem(tw);
  {
    DevMem tmp((size_t)ncs * n);
    ntt_inverse_then_lde(tables_, cs_vals_.p, n, tmp.p, n, cs_coeffs_.p, n, cs_lde_.p, big(), c_.degree_bits, c_.cfg.rate_bits, ncs,
                         gl::GENERATOR, stream_);
    launch_merkle_tree(cs_lde_.p, big(), ncs, big(), c_.cfg.cap_height, cs_tree_.p, stream_);
    P25_HIP(hipStreamSynchronize(stream_));
  }
  const size_t capw = (size_t)4 << c_.cfg.cap_height;
  cs_cap_.resize(capw);
  P25_HIP(hipMemcpy(cs_cap_.data(), cs_tree_.p + tw - capw, capw * 8, hipMemcpyDeviceToHost));
  // circuit_digest = H(cap || H_pad([]) || [degree_bits])  (host; once per circuit)
  {
    std::vector<u64> parts(cs_cap_);
    u64 pad[8] = {1, 0, 0, 0, 0, 0, 0, 1}, ds[4];
    poseidon::hash_no_pad_strided(pad, 1, 8, ds);
    parts.insert(parts.end(), ds, ds + 4);
    parts.push_back((u64)c_.degree_bits);
    poseidon::hash_no_pad_strided(parts.data(), 1, (int)parts.size(), digest_);
  }
  // transcript preamble: circuit_digest | public_inputs_hash (= hash_no_pad([]) = zeros)
  preamble_ = DevMem(8);
  {
    u64 pre[8] = {digest_[0], digest_[1], digest_[2], digest_[3], 0, 0, 0, 0};
    P25_HIP(hipMemcpy(preamble_.p, pre, 64, hipMemcpyHostToDevice));
  }
  k_is_ = DevMem(c_.k_is.size());
  P25_HIP(hipMemcpy(k_is_.p, c_.k_is.data(), c_.k_is.size() * 8, hipMemcpyHostToDevice));

  // quotient argument prototype
  memset(&qa_proto_, 0, sizeof(qa_proto_));
  qa_proto_.cs_lde = cs_lde_.p;
  qa_proto_.pow_big = tables_.pow_table(c_.degree_bits + c_.cfg.rate_bits, false);
  qa_proto_.k_is = k_is_.p;
  qa_proto_.n_gates = (uint32_t)c_.gates.size();
  for (size_t i = 0; i < c_.gates.size(); i++) {
    int s = c_.selector_index[i];
    qa_proto_.gates[i] = GateEntry{(uint32_t)c_.gates[i], (uint32_t)s, (uint32_t)c_.groups[s].first, (uint32_t)c_.groups[s].second};
  }
  qa_proto_.num_selectors = c_.num_selectors;
  qa_proto_.num_wires = c_.cfg.num_wires;
  qa_proto_.num_routed = c_.cfg.num_routed_wires;
  qa_proto_.num_partial_products = c_.num_partial_products;
  qa_proto_.quotient_degree_factor = c_.cfg.max_quotient_degree_factor;
  qa_proto_.degree_bits = c_.degree_bits;
  qa_proto_.rate_bits = c_.cfg.rate_bits;
  {
    u64 g_pow_n = gl::exp_pow2(gl::GENERATOR, c_.degree_bits);
    u64 wr = gl::root_of_unity(c_.cfg.rate_bits);
    for (int k = 0; k < (1 << c_.cfg.rate_bits); k++) {
      qa_proto_.zh[k] = gl::sub(gl::mul(g_pow_n, gl::pow(wr, k)), 1);
      qa_proto_.zh_inv[k] = gl::inv(qa_proto_.zh[k]);
    }
  }
  l0_inv_ = DevMem(big());
  launch_l0_inv(qa_proto_.pow_big, c_.degree_bits, c_.cfg.rate_bits, l0_inv_.p, stream_);
  P25_HIP(hipStreamSynchronize(stream_));
  qa_proto_.l0_inv = l0_inv_.p;
}

DeviceCircuit::~DeviceCircuit() {
  // the proving streams outlive the circuit (pool): nothing of it may still be queued when its buffers go
  for (auto& c : ctxs_)
    if (c->st) (void)hipStreamSynchronize(c->st);
  if (stream_) (void)hipStreamSynchronize(stream_);
  ctxs_.clear();
  for (auto& e : ev_witness_)
    if (e) (void)hipEventDestroy(e);
  for (auto& v : marks_)
    for (auto& e : v) (void)hipEventDestroy(e);
  if (ev_ext_) (void)hipEventDestroy(ev_ext_);
  if (ev_main_) (void)hipEventDestroy(ev_main_);
  for (auto* v : {&kstats_pending_, &kstats_free_})
    for (auto& pr : *v) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
}

void DeviceCircuit::commitment_to_host(std::vector<u64>& coeffs, std::vector<u64>& lde, std::vector<u64>& tree) {
  sync();
  coeffs.resize(cs_coeffs_.words);
  lde.resize(cs_lde_.words);
  tree.resize(cs_tree_.words);
  P25_HIP(hipMemcpy(coeffs.data(), cs_coeffs_.p, coeffs.size() * 8, hipMemcpyDeviceToHost));
  P25_HIP(hipMemcpy(lde.data(), cs_lde_.p, lde.size() * 8, hipMemcpyDeviceToHost));
  P25_HIP(hipMemcpy(tree.data(), cs_tree_.p, tree.size() * 8, hipMemcpyDeviceToHost));
}

void DeviceCircuit::ensure_ctx(size_t count) {
  for (auto& e : ev_witness_)
    if (!e) P25_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  while (ctxs_.size() < count) {
  ctxs_.emplace_back(new Ctx());
  Ctx& x = *ctxs_.back();
  if (ctxs_.size() == 1) pool_first_ = g_stream_pool.reserve(count, (size_t)streams_);
  else g_stream_pool.widen((size_t)streams_);
  x.st = g_stream_pool.at(pool_first_ + ctxs_.size() - 1);
  for (auto& e : x.done) P25_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const size_t n = c_.degree(), B = big();
  const int W = c_.cfg.num_wires, NC = c_.cfg.num_challenges, NP = c_.num_partial_products;
  const int nz = NC * (1 + NP), nq = NC * c_.cfg.max_quotient_degree_factor;
  const size_t tw = merkle_tree_words(B, c_.cfg.cap_height);
  x.wires_vals = DevMem((size_t)W * n);
  x.wires_coeffs = DevMem((size_t)W * n);
  x.tmp = DevMem((size_t)W * n);
  x.wires_lde = DevMem((size_t)W * B);
  x.wires_tree = DevMem(tw);
  x.zs_vals = DevMem((size_t)nz * n);
  x.zs_coeffs = DevMem((size_t)nz * n);
  x.zs_lde = DevMem((size_t)nz * B);
  x.zs_tree = DevMem(tw);
  x.zpp_chunk = DevMem((size_t)NC * (NP + 1) * n);
  x.zpp_tot = DevMem((size_t)NC * n);
  x.zpp_btot = DevMem((size_t)NC * ((n + 255) / 256) + 16);
  x.q_vals = DevMem((size_t)NC * B);
  x.q_tmp = DevMem((size_t)NC * B);
  x.q_coeffs = DevMem((size_t)NC * B);
  x.q_lde = DevMem((size_t)nq * B);
  x.q_tree = DevMem(tw);
  x.preamble = DevMem(8);
  P25_HIP(hipMemcpy(x.preamble.p, preamble_.p, 64, hipMemcpyDeviceToDevice));  // digest | hash_no_pad([]) = zeros
  x.transcript = DevMem(sizeof(Transcript) / 8 + 1);
  x.chal = DevMem(CH_WORDS);
  x.alpha_pows = DevMem(2 * ALPHA_POWS);
  {
    // z^t, t <= 1024, followed by the per-chunk partial sums of the widest oracle (launch_eval_polys)
    const size_t chunks = c_.degree_bits > 16 ? ((size_t)1 << (c_.degree_bits - 16)) : 1;
    size_t widest = std::max<size_t>({(size_t)layout_.oracle_width[0], (size_t)W, (size_t)nz, (size_t)nq});
    x.eval_pows = DevMem(2 * 1026 + 2 * widest * chunks);
  }
  x.fri_comp = DevMem(4 * n);
  size_t total_polys = layout_.oracle_width[0] + layout_.oracle_width[1] + layout_.oracle_width[2] + layout_.oracle_width[3];
  x.fri_scan = DevMem(2 * (total_polys + 1) + 8 * (n + 1) + 4 * ((n + 255) / 256) + 64);
  x.fri.alloc(c_.degree_bits, c_.cfg.rate_bits, c_.cfg.cap_height, c_.fri_reduction_arity_bits);
  x.proof = DevMem(layout_.total);
  x.status = DevMem(1);
  for (auto& e : x.ev) P25_HIP(hipEventCreate(&e));
  x.have_events = true;
  }
}

void DeviceCircuit::ensure_vals(int buf, size_t batch) {
  if (batch <= vals_batch_[buf]) return;
  sync();  // nothing in flight may still read the old allocation
  vals_[buf] = DevMem();
  vals_[buf] = DevMem((size_t)wp_.num_slots * batch);
  vals_batch_[buf] = batch;
}

void DeviceCircuit::sync() {
  P25_HIP(hipStreamSynchronize(stream_));
  for (auto& c : ctxs_) P25_HIP(hipStreamSynchronize(c->st));
}

void DeviceCircuit::stream_join(hipStream_t ext) {
  // one event per proving stream: a context's `join` event is re-recorded here, after the stream's latest work
  if (!ev_main_) P25_HIP(hipEventCreateWithFlags(&ev_main_, hipEventDisableTiming));
  P25_HIP(hipEventRecord(ev_main_, stream_));
  P25_HIP(hipStreamWaitEvent(ext, ev_main_, 0));
  for (auto& c : ctxs_) {
    if (!c->join) P25_HIP(hipEventCreateWithFlags(&c->join, hipEventDisableTiming));
    P25_HIP(hipEventRecord(c->join, c->st));
    P25_HIP(hipStreamWaitEvent(ext, c->join, 0));
  }
}
void DeviceCircuit::wait_stream(hipStream_t ext) {
  // the main stream waits; every proving stream waits for a witness event the main stream records AFTER this point
  // before it touches a proof of a later call
  if (!ev_ext_) P25_HIP(hipEventCreateWithFlags(&ev_ext_, hipEventDisableTiming));
  P25_HIP(hipEventRecord(ev_ext_, ext));
  P25_HIP(hipStreamWaitEvent(stream_, ev_ext_, 0));
}

void DeviceCircuit::mark(int slot) {
  if (slot < 0 || slot >= MAX_MARKS) throw std::invalid_argument("mark slot out of range");
  auto& ev = marks_[slot];
  while (ev.size() < 1 + ctxs_.size()) {
    hipEvent_t e = nullptr;
    P25_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ev.push_back(e);
  }
  P25_HIP(hipEventRecord(ev[0], stream_));
  for (size_t k = 0; k < ctxs_.size(); k++) P25_HIP(hipEventRecord(ev[1 + k], ctxs_[k]->st));
  marks_recorded_[slot] = 1 + ctxs_.size();
}
void DeviceCircuit::stream_wait_mark(hipStream_t ext, int slot) {
  if (slot < 0 || slot >= MAX_MARKS) throw std::invalid_argument("mark slot out of range");
  for (size_t k = 0; k < marks_recorded_[slot]; k++) P25_HIP(hipStreamWaitEvent(ext, marks_[slot][k], 0));
}
void DeviceCircuit::wait_mark(DeviceCircuit& producer, int slot) {
  if (slot < 0 || slot >= MAX_MARKS) throw std::invalid_argument("mark slot out of range");
  for (size_t k = 0; k < producer.marks_recorded_[slot]; k++)
    P25_HIP(hipStreamWaitEvent(stream_, producer.marks_[slot][k], 0));
}

void DeviceCircuit::kernel_stats(double* ms, u64* launches, bool reset) {
  sync();
  for (auto& pr : kstats_pending_) {
    float t = 0;
    P25_HIP(hipEventElapsedTime(&t, pr.first, pr.second));
    kstats_ms_ += t;
    kstats_launches_++;
    kstats_free_.push_back(pr);
  }
  kstats_pending_.clear();
  if (ms) *ms = kstats_ms_;
  if (launches) *launches = kstats_launches_;
  if (reset) {
    kstats_ms_ = 0;
    kstats_launches_ = 0;
  }
}

// One proof, fully enqueued on the stream; no host synchronisation inside.
void DeviceCircuit::prove_one(Ctx& x, int buf, size_t Bstride, uint32_t p, u64* d_proof,
                              uint32_t* d_status, PhaseTimes* t) {
  const u64* d_vals = vals_[buf].p;
  hipStream_t st = x.st;
  const size_t n = c_.degree(), B = big();
  const int W = c_.cfg.num_wires, NC = c_.cfg.num_challenges, NP = c_.num_partial_products;
  const int nz = NC * (1 + NP), nq = NC * c_.cfg.max_quotient_degree_factor;
  const int db = c_.degree_bits, rb = c_.cfg.rate_bits;
  const unsigned cap_h = c_.cfg.cap_height;
  const size_t capw = (size_t)4 << cap_h;
  const size_t tw = merkle_tree_words(B, cap_h);
  const ProofLayout& L = layout_;
  Transcript* tr = (Transcript*)x.transcript.p;
  u64* chal = x.chal.p;
  int evi = 0;
  auto mark = [&]() {
    if (t) P25_HIP(hipEventRecord(x.ev[evi++], st));
  };
  auto d2d = [&](u64* dst, const u64* src, size_t words) {
    P25_HIP(hipMemcpyAsync(dst, src, words * 8, hipMemcpyDeviceToDevice, st));
  };
  mark();  // 0
  // "compute full witness" + "compute wire polynomials"
  launch_fill_wires(wp_, d_vals, Bstride, p, x.wires_vals.p, st);
  // "let public_inputs = partition_witness.get_targets(&prover_data.public_inputs); let public_inputs_hash = ..."
  // (no launch at all for a circuit without public inputs: the preamble keeps the hash of the empty list)
  if (wp_.n_public_inputs)
    launch_public_inputs(d_vals, Bstride, p, wp_.d_pi_slots, wp_.n_public_inputs, d_proof + L.public_inputs, x.preamble.p + 4, st);
  P25_HIP(hipEventRecord(x.done[buf], st));  // last read of this witness-value buffer by this proof
  mark();  // 1
  // "compute wires commitment"
  ntt_inverse_then_lde(tables_, x.wires_vals.p, n, x.tmp.p, n, x.wires_coeffs.p, n, x.wires_lde.p, B, db, rb, W, gl::GENERATOR, st);
  {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (kstats_on_) {
      if (kstats_free_.empty()) {
        P25_HIP(hipEventCreate(&e0));
        P25_HIP(hipEventCreate(&e1));
      } else {
        e0 = kstats_free_.back().first;
        e1 = kstats_free_.back().second;
        kstats_free_.pop_back();
      }
      kstats_pending_.push_back({e0, e1});
    }
    launch_merkle_tree(x.wires_lde.p, B, W, B, cap_h, x.wires_tree.p, st, e0, e1, single_proof_);
  }
  const u64* wires_cap = x.wires_tree.p + tw - capw;
  d2d(d_proof + L.wires_cap, wires_cap, capw);
  launch_transcript(tr, 1, x.preamble.p, 8, chal, 0, st);
  launch_transcript(tr, 0, wires_cap, (uint32_t)capw, chal + CH_BETAS, 2 * NC, st);  // betas, gammas
  mark();  // 2
  // "compute partial products"
  enqueue_partial_products(x, st);
  mark();  // 3
  // "commit to partial products, Z's"
  ntt_inverse_then_lde(tables_, x.zs_vals.p, n, x.tmp.p, n, x.zs_coeffs.p, n, x.zs_lde.p, B, db, rb, nz, gl::GENERATOR, st);
  launch_merkle_tree(x.zs_lde.p, B, nz, B, cap_h, x.zs_tree.p, st, nullptr, nullptr, single_proof_);
  const u64* zs_cap = x.zs_tree.p + tw - capw;
  d2d(d_proof + L.zs_cap, zs_cap, capw);
  launch_transcript(tr, 0, zs_cap, (uint32_t)capw, chal + CH_ALPHAS, NC, st);
  mark();  // 4
  // "compute quotient polys"
  enqueue_quotient(x, st);
  mark();  // 5
  // "split up quotient polys" (chunks of n are contiguous: [NC][8][n] == [16][n]) + "commit to quotient polys"
  ntt_lde_bitrev(tables_, x.q_coeffs.p, n, x.q_lde.p, B, db, rb, nq, gl::GENERATOR, st);
  launch_merkle_tree(x.q_lde.p, B, nq, B, cap_h, x.q_tree.p, st, nullptr, nullptr, single_proof_);
  const u64* q_cap = x.q_tree.p + tw - capw;
  d2d(d_proof + L.quotient_cap, q_cap, capw);
  launch_transcript(tr, 0, q_cap, (uint32_t)capw, chal + CH_ZETA, 2, st);
  hipLaunchKernelGGL(k_check_zeta, dim3(1), dim3(1), 0, st, chal, (uint32_t)db, d_status);
  mark();  // 6
  // "construct the opening set"
  {
    const u64 g = gl::root_of_unity(db);
    launch_eval_polys(cs_coeffs_.p, L.oracle_width[0], db, chal + CH_ZETA, 1, x.eval_pows.p, d_proof + L.constants, st);
    launch_eval_polys(x.wires_coeffs.p, W, db, chal + CH_ZETA, 1, x.eval_pows.p, d_proof + L.wires, st, true);
    launch_eval_polys(x.zs_coeffs.p, NC, db, chal + CH_ZETA, 1, x.eval_pows.p, d_proof + L.zs, st, true);
    launch_eval_polys(x.zs_coeffs.p + (size_t)NC * n, NC * NP, db, chal + CH_ZETA, 1, x.eval_pows.p, d_proof + L.pps, st, true);
    launch_eval_polys(x.q_coeffs.p, nq, db, chal + CH_ZETA, 1, x.eval_pows.p, d_proof + L.quotient, st, true);
    launch_eval_polys(x.zs_coeffs.p, NC, db, chal + CH_ZETA, g, x.eval_pows.p, d_proof + L.zs_next, st);
    // observe: constants|sigmas|wires|zs, then pps|quotient, then zs_next; then FRI alpha
    launch_transcript(tr, 0, d_proof + L.constants, (uint32_t)(L.zs_next - L.constants), chal, 0, st);
    launch_transcript(tr, 0, d_proof + L.pps, (uint32_t)(L.fri_caps - L.pps), chal, 0, st);
    launch_transcript(tr, 0, d_proof + L.zs_next, (uint32_t)(L.pps - L.zs_next), chal + CH_FRI_ALPHA, 2, st);
    mark();  // 7
    // "compute opening proofs": batch, divide, multiply by X
    FriCombineArgs fa;
    fa.coeffs[0] = cs_coeffs_.p;
    fa.coeffs[1] = x.wires_coeffs.p;
    fa.coeffs[2] = x.zs_coeffs.p;
    fa.coeffs[3] = x.q_coeffs.p;
    for (int k = 0; k < 4; k++) fa.n_polys[k] = L.oracle_width[k];
    fa.log_n = db;
    fa.num_challenges = NC;
    fa.chal = chal;
    fa.g = g;
    fa.comp = x.fri_comp.p;
    fa.scan_tmp = x.fri_scan.p;
    fa.final_a = x.fri.coeffs[0].p;
    fa.final_b = x.fri.coeffs[0].p + n;
    launch_fri_combine(fa, st);
  }
  // commit phase, PoW, query rounds
  {
    QueryArgs qy;
    memset(&qy, 0, sizeof(qy));
    const u64* ldes[4] = {cs_lde_.p, x.wires_lde.p, x.zs_lde.p, x.q_lde.p};
    const u64* trees[4] = {cs_tree_.p, x.wires_tree.p, x.zs_tree.p, x.q_tree.p};
    qy.n_oracles = 4;
    for (int k = 0; k < 4; k++) {
      qy.oracle_lde[k] = ldes[k];
      qy.oracle_tree[k] = trees[k];
      qy.oracle_width[k] = L.oracle_width[k];
    }
    FriShape sh{db, rb, cap_h, c_.fri_reduction_arity_bits, c_.cfg.proof_of_work_bits, c_.cfg.num_query_rounds};
    FriOffsets fo{L.fri_caps, L.final_poly, L.pow_witness, L.queries, L.query_stride};
    fri_commit_pow_query(tables_, x.fri, sh, tr, chal, qy, d_proof, fo, d_status, st, single_proof_);
  }
  mark();  // 8
  P25_HIP(hipGetLastError());
  if (t) {
    P25_HIP(hipEventSynchronize(x.ev[8]));
    float ms[8];
    for (int i = 0; i < 8; i++) P25_HIP(hipEventElapsedTime(&ms[i], x.ev[i], x.ev[i + 1]));
    t->witness += ms[0];
    t->wires_commit += ms[1];
    t->zs_pp += ms[2];
    t->zs_commit += ms[3];
    t->quotient += ms[4];
    t->quotient_commit += ms[5];
    t->openings += ms[6];
    t->fri += ms[7];
    float tot;
    P25_HIP(hipEventElapsedTime(&tot, x.ev[0], x.ev[8]));
    t->total += tot;
  }
}

// "compute partial products": Z and partial products (values, natural row order) from the witness values and
// betas / gammas of the context's challenge block.
void DeviceCircuit::enqueue_partial_products(Ctx& x, hipStream_t st) {
  const size_t n = c_.degree();
  const int NC = c_.cfg.num_challenges, NP = c_.num_partial_products, db = c_.degree_bits;
  u64* chal = x.chal.p;
  {
    ZppArgs za;
    za.wires = x.wires_vals.p;
    za.sigmas = cs_vals_.p + (size_t)(c_.num_selectors + c_.cfg.num_constants) * n;
    za.pow_n = tables_.pow_table(db, false);
    za.k_is = k_is_.p;
    za.chal = chal;
    za.chunk = x.zpp_chunk.p;
    za.tot = x.zpp_tot.p;
    za.block_tot = x.zpp_btot.p;
    za.out = x.zs_vals.p;
    za.n = (uint32_t)n;
    za.num_routed = c_.cfg.num_routed_wires;
    za.num_partial_products = NP;
    za.num_challenges = NC;
    za.quotient_degree_factor = c_.cfg.max_quotient_degree_factor;
    launch_zpp(za, st);
  }
}

// "compute quotient polys": quotient values on the coset from the wires / Z LDEs and the challenge block, then the
// coset iNTT to 8n coefficients per challenge (= the NC x 8 chunks of n, contiguous).
void DeviceCircuit::enqueue_quotient(Ctx& x, hipStream_t st) {
  const size_t B = big();
  const int NC = c_.cfg.num_challenges, lde_bits = c_.degree_bits + c_.cfg.rate_bits;
  u64* chal = x.chal.p;
  launch_alpha_pows(chal, x.alpha_pows.p, st);
  {
    QuotientArgs qa = qa_proto_;
    qa.wires_lde = x.wires_lde.p;
    qa.zs_lde = x.zs_lde.p;
    qa.chal = chal;
    qa.pi_hash = x.preamble.p + 4;
    qa.alpha_pows = x.alpha_pows.p;
    qa.out = x.q_vals.p;
    launch_quotient(qa, st);
  }
  // coset iNTT of the 2 value vectors (stored at bit-reversed positions) -> 8n coefficients each
  ntt_inverse(tables_, x.q_vals.p, B, true, x.q_tmp.p, B, x.q_coeffs.p, B, lde_bits, NC, gl::GENERATOR, st);
}

// ---------------------------------------------------------------- isolated stages (parity tests; include/p25.h)
void DeviceCircuit::set_challenges(Ctx& x, const u64* betas, const u64* gammas, const u64* alphas) {
  u64 ch[CH_WORDS] = {0};
  for (int i = 0; i < c_.cfg.num_challenges; i++) {
    if (betas) ch[CH_BETAS + i] = betas[i];
    if (gammas) ch[CH_GAMMAS + i] = gammas[i];
    if (alphas) ch[CH_ALPHAS + i] = alphas[i];
  }
  for (u64 v : ch)
    if (v >= gl::P) throw std::invalid_argument("non-canonical challenge");
  P25_HIP(hipMemcpyAsync(x.chal.p, ch, sizeof(ch), hipMemcpyHostToDevice, x.st));
  P25_HIP(hipStreamSynchronize(x.st));  // ch is on this stack
}
static void check_canonical(const u64* v, size_t n, const char* what) {
  for (size_t i = 0; i < n; i++)
    if (v[i] >= gl::P) throw std::invalid_argument(std::string("non-canonical field element in ") + what);
}
void DeviceCircuit::partial_products(const u64* wires, const u64* betas, const u64* gammas, u64* out) {
  ensure_ctx(1);
  sync();
  Ctx& x = *ctxs_[0];
  const size_t n = c_.degree();
  const size_t W = c_.cfg.num_wires, nz = (size_t)c_.cfg.num_challenges * (1 + c_.num_partial_products);
  check_canonical(wires, W * n, "wires");
  set_challenges(x, betas, gammas, nullptr);
  P25_HIP(hipMemcpyAsync(x.wires_vals.p, wires, W * n * 8, hipMemcpyHostToDevice, x.st));
  enqueue_partial_products(x, x.st);
  P25_HIP(hipMemcpyAsync(out, x.zs_vals.p, nz * n * 8, hipMemcpyDeviceToHost, x.st));
  P25_HIP(hipStreamSynchronize(x.st));
  P25_HIP(hipGetLastError());
}
void DeviceCircuit::quotient(const u64* wires, const u64* zs_pp, const u64* betas, const u64* gammas, const u64* alphas,
                             u64* out) {
  ensure_ctx(1);
  sync();
  Ctx& x = *ctxs_[0];
  const size_t n = c_.degree(), B = big();
  const int W = c_.cfg.num_wires, NC = c_.cfg.num_challenges, nz = NC * (1 + c_.num_partial_products);
  const int db = c_.degree_bits, rb = c_.cfg.rate_bits;
  check_canonical(wires, (size_t)W * n, "wires");
  check_canonical(zs_pp, (size_t)nz * n, "zs_partial_products");
  set_challenges(x, betas, gammas, alphas);
  hipStream_t st = x.st;
  if (c_.pi_row >= 0) {  // the public-inputs hash the PublicInputGate compares with is what that row of the witness holds
    u64 pih[4];
    for (int i = 0; i < 4; i++) pih[i] = wires[(size_t)i * n + (size_t)c_.pi_row];
    P25_HIP(hipMemcpyAsync(x.preamble.p + 4, pih, 32, hipMemcpyHostToDevice, st));
    P25_HIP(hipStreamSynchronize(st));  // pih is on this stack
  }
  P25_HIP(hipMemcpyAsync(x.wires_vals.p, wires, (size_t)W * n * 8, hipMemcpyHostToDevice, st));
  P25_HIP(hipMemcpyAsync(x.zs_vals.p, zs_pp, (size_t)nz * n * 8, hipMemcpyHostToDevice, st));
  ntt_inverse_then_lde(tables_, x.wires_vals.p, n, x.tmp.p, n, x.wires_coeffs.p, n, x.wires_lde.p, B, db, rb, W, gl::GENERATOR, st);
  ntt_inverse_then_lde(tables_, x.zs_vals.p, n, x.tmp.p, n, x.zs_coeffs.p, n, x.zs_lde.p, B, db, rb, nz, gl::GENERATOR, st);
  enqueue_quotient(x, st);
  P25_HIP(hipMemcpyAsync(out, x.q_coeffs.p, (size_t)NC * B * 8, hipMemcpyDeviceToHost, st));
  // context 0 goes back to proving: a circuit without registered public inputs never rewrites the hash words, so put
  // hash_no_pad([]) back instead of leaving the caller's (possibly arbitrary) wires there
  if (c_.pi_row >= 0) P25_HIP(hipMemcpyAsync(x.preamble.p + 4, preamble_.p + 4, 32, hipMemcpyDeviceToDevice, st));
  P25_HIP(hipStreamSynchronize(st));
  P25_HIP(hipGetLastError());
}

// Challenger script on the device transcript (upstream iop/challenger.rs): per segment observe, then draw.
void transcript_script(const u64* obs, const uint32_t* seg_len, const uint32_t* n_chal, size_t n_seg, u64* out) {
  size_t n_obs = 0, n_out = 0;
  for (size_t k = 0; k < n_seg; k++) {
    if (n_chal[k] > 64 || seg_len[k] > (1u << 24)) throw std::invalid_argument("transcript segment too large");
    n_obs += seg_len[k];
    n_out += n_chal[k];
  }
  check_canonical(obs, n_obs, "observed words");
  DevMem d_tr(sizeof(Transcript) / 8 + 1), d_obs(n_obs + 1), d_out(n_out + 1);
  P25_HIP(hipMemcpy(d_obs.p, obs, n_obs * 8, hipMemcpyHostToDevice));
  size_t o = 0, c = 0;
  for (size_t k = 0; k < n_seg; k++) {
    launch_transcript((Transcript*)d_tr.p, k == 0, d_obs.p + o, seg_len[k], d_out.p + c, n_chal[k], 0);
    o += seg_len[k];
    c += n_chal[k];
  }
  if (!n_seg) return;
  P25_HIP(hipMemcpy(out, d_out.p, n_out * 8, hipMemcpyDeviceToHost));
  P25_HIP(hipGetLastError());
}

size_t fri_prove_words(const FriShape& sh) {
  const size_t capw = (size_t)4 << sh.cap_height, nl = sh.arity_bits.size();
  int deg = sh.log_n, bits = sh.log_n + sh.rate_bits;
  size_t per_q = 0;
  for (int a : sh.arity_bits) {
    deg -= a;
    bits -= a;
    per_q += 2 * ((size_t)1 << a) + 4 * (size_t)(bits - (int)sh.cap_height);
  }
  return nl * capw + 2 * nl + 2 * ((size_t)1 << deg) + 1 + sh.num_queries + per_q * sh.num_queries;
}
// out: CAP[n_layers] | betas E[n_layers] | final_poly E[..] | pow_witness | indices[num_queries] | query openings
void fri_prove_standalone(NttTables& tables, const u64* coeffs, const FriShape& sh, const u64* seed, size_t n_seed,
                          u64* out, int32_t* status_out) {
  const size_t n = (size_t)1 << sh.log_n, capw = (size_t)4 << sh.cap_height, nl = sh.arity_bits.size();
  check_canonical(coeffs, 2 * n, "polynomial coefficients");
  check_canonical(seed, n_seed, "transcript seed");
  int deg = sh.log_n, bits = sh.log_n + sh.rate_bits;
  size_t per_q = 0;
  for (int a : sh.arity_bits) {
    if (a < 1 || a > 8) throw std::invalid_argument("FRI arity bits must be in 1..8");
    deg -= a;
    bits -= a;
    if (deg < 0 || bits < (int)sh.cap_height) throw std::invalid_argument("FRI layer smaller than the Merkle cap");
    per_q += 2 * ((size_t)1 << a) + 4 * (size_t)(bits - (int)sh.cap_height);
  }
  if (sh.num_queries < 1 || sh.num_queries > 64 || sh.pow_bits < 0 || sh.pow_bits > 32 || sh.rate_bits < 0 || sh.rate_bits > 3)
    throw std::invalid_argument("FRI parameters out of range");
  FriWork w;
  w.alloc(sh.log_n, sh.rate_bits, sh.cap_height, sh.arity_bits);
  FriOffsets fo;
  fo.caps = 0;
  fo.final_poly = nl * capw;
  fo.pow_witness = fo.final_poly + 2 * ((size_t)1 << deg);
  fo.queries = fo.pow_witness + 1;
  fo.query_stride = per_q;
  const size_t proof_words = fo.queries + per_q * sh.num_queries;
  DevMem d_tr(sizeof(Transcript) / 8 + 1), d_chal(CH_WORDS), d_proof(proof_words), d_status(1), d_seed(n_seed + 1);
  hipStream_t st = 0;
  P25_HIP(hipMemcpy(w.coeffs[0].p, coeffs, 2 * n * 8, hipMemcpyHostToDevice));
  P25_HIP(hipMemcpy(d_seed.p, seed, n_seed * 8, hipMemcpyHostToDevice));
  P25_HIP(hipMemset(d_status.p, 0, 8));
  P25_HIP(hipMemset(d_chal.p, 0, CH_WORDS * 8));
  launch_transcript((Transcript*)d_tr.p, 1, d_seed.p, (uint32_t)n_seed, d_chal.p, 0, st);
  QueryArgs qy;
  memset(&qy, 0, sizeof(qy));
  fri_commit_pow_query(tables, w, sh, (Transcript*)d_tr.p, d_chal.p, qy, d_proof.p, fo, (uint32_t*)d_status.p, st, true);
  P25_HIP(hipStreamSynchronize(st));
  P25_HIP(hipGetLastError());
  std::vector<u64> pr(proof_words), ch(CH_WORDS);
  uint32_t hs[2] = {0, 0};
  P25_HIP(hipMemcpy(pr.data(), d_proof.p, proof_words * 8, hipMemcpyDeviceToHost));
  P25_HIP(hipMemcpy(ch.data(), d_chal.p, CH_WORDS * 8, hipMemcpyDeviceToHost));
  P25_HIP(hipMemcpy(hs, d_status.p, 8, hipMemcpyDeviceToHost));
  *status_out = (int32_t)hs[0];
  u64* o = out;
  memcpy(o, pr.data(), nl * capw * 8); o += nl * capw;
  memcpy(o, ch.data() + CH_FRI_BETAS, 2 * nl * 8); o += 2 * nl;
  memcpy(o, pr.data() + fo.final_poly, (fo.queries - fo.final_poly) * 8); o += fo.queries - fo.final_poly;
  const u64 mask = ((u64)1 << (sh.log_n + sh.rate_bits)) - 1;
  for (int q = 0; q < sh.num_queries; q++) *o++ = ch[CH_QUERIES + q] & mask;
  memcpy(o, pr.data() + fo.queries, per_q * sh.num_queries * 8);
}

// Batch schedule: witness generation for up to 64 proofs at a time on the main stream (it is
// parallel ACROSS proofs), then each proof's commit/quotient/FRI pipeline on one of K streams so that
// the latency-bound stretches of one proof (transcript, Merkle-cap levels, FRI tail) overlap with the
// throughput-bound kernels of the others.
// Approximate device bytes of one proof context (see ensure_ctx): the LDE matrices dominate.
size_t DeviceCircuit::ctx_bytes() const {
  const size_t n = c_.degree(), B = (size_t)1 << (c_.degree_bits + c_.cfg.rate_bits);
  const size_t W = c_.cfg.num_wires, NC = c_.cfg.num_challenges, NP = c_.num_partial_products;
  const size_t nz = NC * (1 + NP), nq = NC * c_.cfg.max_quotient_degree_factor;
  const size_t tw = merkle_tree_words(B, c_.cfg.cap_height);
  return 8 * (3 * W * n + W * B + 2 * nz * n + nz * B + 3 * NC * B + nq * B + 3 * tw + 16 * n + 6 * B);
}

void DeviceCircuit::prove_batch_dev(const u64* d_inputs, size_t n_proofs, const u64* d_seeds, u64* d_proofs,
                                    size_t proof_stride, uint32_t* d_status, PhaseTimes* times, const u64* d_filler,
                                    size_t in_stride, size_t in_max_off) {
  size_t K = times ? 1 : (size_t)streams_;
  if (K > n_proofs) K = n_proofs ? n_proofs : 1;
  if (K > ctxs_.size()) {
    // New contexts must fit in what is actually FREE on the device (other circuits, the caller's tensors and
    // this circuit's tables are already allocated), after the witness-value array of this call and a 5%
    // reserve -- matters for 2^19-row circuits (13 GB per context).
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b) {
      const size_t pass = n_proofs < 64 ? n_proofs : 64;
      size_t vals_need = 0;
      for (int b = 0; b < (n_proofs > 64 ? 2 : 1); b++)
        if (pass > vals_batch_[b]) vals_need += (size_t)wp_.num_slots * pass * 8;
      const size_t reserve = total_b / 20;
      const size_t avail = free_b > vals_need + reserve ? free_b - vals_need - reserve : 0;
      size_t fit = ctxs_.size() + avail / ctx_bytes();
      if (fit < 1) fit = 1;
      if (K > fit) K = fit;
    }
  }
  ensure_ctx(K);
  single_proof_ = K == 1;  // a lone proof in flight: latency-oriented kernel forms
  // Witness generation runs for up to 64 proofs per pass on the main stream into one of TWO value buffers, so the
  // pass for proofs [k+64, k+128) runs underneath the proving pipelines of [k, k+64): a context stream only waits for
  // the witness event of its own pass, and the main stream only waits -- before it overwrites buffer b -- for the
  // contexts' latest reads of buffer b (two passes back).  No point of the batch drains the proving streams.
  const size_t MAXB = 64;
  size_t pass = 0;
  for (size_t base = 0; base < n_proofs; base += MAXB, pass++) {
    size_t bsz = n_proofs - base < MAXB ? n_proofs - base : MAXB;
    const int buf = (int)((pass_counter_ + pass) & 1);
    ensure_vals(buf, bsz);
    for (auto& c : ctxs_) P25_HIP(hipStreamWaitEvent(stream_, c->done[buf], 0));
    P25_HIP(hipMemsetAsync(d_status + base, 0, bsz * 4, stream_));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (times) {
      P25_HIP(hipEventCreate(&e0));
      P25_HIP(hipEventCreate(&e1));
      P25_HIP(hipEventRecord(e0, stream_));
    }
    launch_witgen(wp_, d_inputs, d_seeds + base, vals_[buf].p, bsz, (uint32_t)bsz, d_status + base,
                  stream_, d_filler ? d_filler + base * wp_.num_random_fill : nullptr, in_stride, in_max_off, base);
    P25_HIP(hipEventRecord(ev_witness_[buf], stream_));
    if (times) {
      P25_HIP(hipEventRecord(e1, stream_));
      P25_HIP(hipEventSynchronize(e1));
      float ms;
      P25_HIP(hipEventElapsedTime(&ms, e0, e1));
      times->witness += ms;
      times->total += ms;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
    for (size_t k = 0; k < K && k < bsz; k++) P25_HIP(hipStreamWaitEvent(ctxs_[k]->st, ev_witness_[buf], 0));
    for (size_t p = 0; p < bsz; p++)
      prove_one(*ctxs_[p % K], buf, bsz, (uint32_t)p, d_proofs + (base + p) * proof_stride, d_status + base + p, times);
  }
  pass_counter_ += pass;
  P25_HIP(hipGetLastError());
}

void DeviceCircuit::prove_batch(const u64* inputs, size_t n_proofs, const u64* seeds, u64* proofs_out,
                                size_t proof_stride, int32_t* statuses, PhaseTimes* times, const u64* filler) {
  if (proof_stride < layout_.total) throw std::invalid_argument("proof_stride smaller than the proof");
  const size_t ni = wp_.n_inputs;
  for (size_t i = 0; i < n_proofs * ni; i++)
    if (inputs[i] >= gl::P) throw std::invalid_argument("non-canonical input field element");
  DevMem d_in(n_proofs * ni), d_seeds(n_proofs), d_proofs(n_proofs * layout_.total), d_status((n_proofs + 1) / 2 + 1);
  std::vector<u64> sd(n_proofs);
  for (size_t i = 0; i < n_proofs; i++) sd[i] = seeds ? seeds[i] : (u64)i;
  P25_HIP(hipMemcpyAsync(d_in.p, inputs, n_proofs * ni * 8, hipMemcpyHostToDevice, stream_));
  P25_HIP(hipMemcpyAsync(d_seeds.p, sd.data(), n_proofs * 8, hipMemcpyHostToDevice, stream_));
  DevMem d_filler(filler ? n_proofs * wp_.num_random_fill + 1 : 0);
  if (filler) {
    for (size_t i = 0; i < n_proofs * wp_.num_random_fill; i++)
      if (filler[i] >= gl::P) throw std::invalid_argument("non-canonical filler field element");
    P25_HIP(hipMemcpyAsync(d_filler.p, filler, n_proofs * wp_.num_random_fill * 8, hipMemcpyHostToDevice, stream_));
  }
  prove_batch_dev(d_in.p, n_proofs, d_seeds.p, d_proofs.p, layout_.total, (uint32_t*)d_status.p, times, filler ? d_filler.p : nullptr);
  sync();
  std::vector<uint32_t> hs(n_proofs);
  P25_HIP(hipMemcpyAsync(hs.data(), d_status.p, n_proofs * 4, hipMemcpyDeviceToHost, stream_));
  for (size_t i = 0; i < n_proofs; i++)
    P25_HIP(hipMemcpyAsync(proofs_out + i * proof_stride, d_proofs.p + i * layout_.total, layout_.total * 8,
                           hipMemcpyDeviceToHost, stream_));
  P25_HIP(hipStreamSynchronize(stream_));
  for (size_t i = 0; i < n_proofs; i++) statuses[i] = (int32_t)hs[i];
}

int32_t DeviceCircuit::witness(const u64* inputs, u64 seed, u64* wires_out) {
  ensure_ctx(1);
  sync();
  ensure_vals(0, 1);
  Ctx* ctx_ = ctxs_[0].get();
  const size_t ni = wp_.n_inputs;
  for (size_t i = 0; i < ni; i++)
    if (inputs[i] >= gl::P) throw std::invalid_argument("non-canonical input field element");
  DevMem d_in(ni), d_seed(1);
  P25_HIP(hipMemcpyAsync(d_in.p, inputs, ni * 8, hipMemcpyHostToDevice, stream_));
  P25_HIP(hipMemcpyAsync(d_seed.p, &seed, 8, hipMemcpyHostToDevice, stream_));
  uint32_t* d_status = (uint32_t*)ctx_->status.p;
  P25_HIP(hipMemsetAsync(d_status, 0, 4, stream_));
  launch_witgen(wp_, d_in.p, d_seed.p, vals_[0].p, 1, 1, d_status, stream_);
  launch_fill_wires(wp_, vals_[0].p, 1, 0, ctx_->wires_vals.p, stream_);
  uint32_t hs = 0;
  P25_HIP(hipMemcpyAsync(&hs, d_status, 4, hipMemcpyDeviceToHost, stream_));
  P25_HIP(hipMemcpyAsync(wires_out, ctx_->wires_vals.p, wp_.n_wire_elems * 8, hipMemcpyDeviceToHost, stream_));
  P25_HIP(hipStreamSynchronize(stream_));
  P25_HIP(hipGetLastError());
  return (int32_t)hs;
}

}  // namespace p25
