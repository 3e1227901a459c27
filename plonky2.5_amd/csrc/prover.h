// Host orchestration of the GPU prover: device-resident circuit tables + per-stream proof contexts.
// Mirrors the phases of upstream plonky2 @ 3de92d9 prover.rs `prove_with_partition_witness`
// (reached from /root/reference/src/p3/mod.rs:260); phase names follow upstream's timing spans
// (SURVEY.md App. A.9).  No host computation on the per-proof path: the host only enqueues kernels.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "builder.h"
#include "prover_kernels.h"

namespace p25 {

struct ProofLayout {
  // word offsets into the flat proof (include/p25.h "Proof layout")
  size_t wires_cap, zs_cap, quotient_cap;
  size_t constants, sigmas, wires, zs, zs_next, pps, quotient;
  size_t fri_caps, queries, query_stride, final_poly, pow_witness, public_inputs, total;
  uint32_t num_public_inputs;
  uint32_t oracle_width[4];
  uint32_t final_poly_len;
};
ProofLayout make_proof_layout(const Circuit& c);

struct DevMem {  // owning device allocation
  u64* p = nullptr;
  size_t words = 0;
  DevMem() {}
  explicit DevMem(size_t w);
  ~DevMem();
  DevMem(DevMem&& o) noexcept : p(o.p), words(o.words) { o.p = nullptr; }
  DevMem& operator=(DevMem&& o) noexcept;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
};

// FRI working set and shape: shared by the whole-proof pipeline and the standalone p25_fri_prove entry point.
struct FriWork {
  DevMem coeffs[9], vals[9], tree[9];
  void alloc(int log_n, int rate_bits, unsigned cap_height, const std::vector<int>& arity_bits);
};
struct FriShape {
  int log_n, rate_bits;
  unsigned cap_height;
  std::vector<int> arity_bits;
  int pow_bits, num_queries;
};
struct FriOffsets {  // word offsets into the flat proof buffer
  size_t caps, final_poly, pow_witness, queries, query_stride;
};
void fri_commit_pow_query(NttTables& tables, FriWork& w, const FriShape& sh, Transcript* tr, u64* chal, QueryArgs qy,
                          u64* d_proof, const FriOffsets& fo, uint32_t* d_status, hipStream_t st, bool single_proof);

void transcript_script(const u64* obs, const uint32_t* seg_len, const uint32_t* n_chal, size_t n_seg, u64* out);
size_t fri_prove_words(const FriShape& sh);
void fri_prove_standalone(NttTables& tables, const u64* coeffs, const FriShape& sh, const u64* seed, size_t n_seed,
                          u64* out, int32_t* status_out);

struct PhaseTimes {  // milliseconds, device time measured with HIP events on the proving stream
  float witness = 0, wires_commit = 0, zs_pp = 0, zs_commit = 0, quotient = 0, quotient_commit = 0, openings = 0,
        fri = 0, total = 0;
};

// A circuit's share of the process-wide pool's two main streams (prover.hip: StreamPool); slot -1 = none held.
struct MainStreamLease {
  int slot = -1, device = 0;
  MainStreamLease() = default;
  MainStreamLease(const MainStreamLease&) = delete;
  MainStreamLease& operator=(const MainStreamLease&) = delete;
  ~MainStreamLease();
};

class DeviceCircuit {
 public:
  explicit DeviceCircuit(Circuit c);
  ~DeviceCircuit();
  const Circuit& circuit() const { return c_; }
  const ProofLayout& layout() const { return layout_; }
  const u64* digest() const { return digest_; }                 // host copy [4]
  const std::vector<u64>& cs_cap() const { return cs_cap_; }    // host copy
  // host copies of the constants/sigmas commitment (coefficients, LDE in leaf order, Merkle tree) for
  // CircuitData::to_bytes (circuit_bytes.h)
  void commitment_to_host(std::vector<u64>& coeffs, std::vector<u64>& lde, std::vector<u64>& tree);
  size_t n() const { return c_.degree(); }
  size_t big() const { return c_.degree() << c_.cfg.rate_bits; }

  // Proves n_proofs independent inputs.  inputs[n_proofs][num_inputs] (host), seeds (host, nullable),
  // proofs_out[n_proofs][proof_stride] (host), statuses[n_proofs].  Returns first non-zero HIP-level
  // failure as exception; per-proof failures go to statuses.
  // filler (nullable): explicit RandomValueGenerator values [n_proofs][num_random_fill()] replacing the seeds
  void prove_batch(const u64* inputs, size_t n_proofs, const u64* seeds, u64* proofs_out, size_t proof_stride,
                   int32_t* statuses, PhaseTimes* times, const u64* filler = nullptr);
  uint32_t num_random_fill() const { return wp_.num_random_fill; }
  // debug / parity: full witness of one input -> wires[num_wires][n] (host)
  int32_t witness(const u64* inputs, u64 seed, u64* wires_out);

  // Device-resident API used by bench.py: inputs already in HBM ([n_proofs][num_inputs]), proofs
  // written to a device buffer; asynchronous on the internal stream until sync().
  // in_stride / in_max_off: proof p reads its inputs at d_inputs + min(p * in_stride, in_max_off) words (0 / -1: the plain array)
  void prove_batch_dev(const u64* d_inputs, size_t n_proofs, const u64* d_seeds, u64* d_proofs, size_t proof_stride,
                       uint32_t* d_status, PhaseTimes* times, const u64* d_filler = nullptr, size_t in_stride = 0,
                       size_t in_max_off = (size_t)-1);
  void sync();
  // Device-side ordering against a caller's stream, no host synchronisation (the multi-GPU gather runs on a side
  // stream underneath the next step's proofs):
  //   stream_join(ext):  `ext` waits for everything this circuit has enqueued so far (all proving streams)
  //   wait_stream(ext):  everything this circuit enqueues from now on waits for what `ext` holds now
  void stream_join(hipStream_t ext);
  void wait_stream(hipStream_t ext);
  // The same between two CIRCUITS, through events only (no helper stream whose wait packets would sit in a hardware
  // queue in front of unrelated work): mark(slot) records the tail of every proving stream of this circuit;
  // wait_mark(producer, slot) makes everything this circuit enqueues from now on wait for the producer's mark.
  // A level of an aggregation tree proves straight on the buffer the level below wrote (plonky25_amd.aggregate).
  static constexpr int MAX_MARKS = 16;   // include/p25.h: P25_MAX_MARKS
  void mark(int slot);
  void wait_mark(DeviceCircuit& producer, int slot);
  void stream_wait_mark(hipStream_t ext, int slot);   // a caller's stream waits for mark(slot) (stream_join, but lag-able)
  // proofs kept in flight by prove_batch* (one HIP stream + working set each), 1..16
  void set_streams(int k) { streams_ = k < 1 ? 1 : (k > 32 ? 32 : k); }
  hipStream_t stream() const { return stream_; }
  // Isolated stages (host buffers; parity tests of SURVEY 8a7 / 8a8 through p25_partial_products / p25_quotient):
  // wires[num_wires][n] -> out[NC*(1+NP)][n];  wires + zs_pp values -> out[NC*8][n] quotient chunk coefficients
  void partial_products(const u64* wires, const u64* betas, const u64* gammas, u64* out);
  void quotient(const u64* wires, const u64* zs_pp, const u64* betas, const u64* gammas, const u64* alphas, u64* out);
  // Dominant-kernel accounting for bench.py's roofline line: HIP events bracket every launch of the
  // wires leaf-sponge kernel (k_hash_leaves over the 135-column LDE) on the proving stream.
  void kernel_stats_enable(bool on) { kstats_on_ = on; }
  // drains finished event pairs; returns accumulated (ms, launches) since the last reset
  void kernel_stats(double* ms, u64* launches, bool reset);

 private:
  struct Ctx;  // per-proof working set
  void prove_one(Ctx& cx, int buf, size_t B, uint32_t p, u64* d_proof, uint32_t* d_status, PhaseTimes* t);
  void enqueue_partial_products(Ctx& cx, hipStream_t st);
  void enqueue_quotient(Ctx& cx, hipStream_t st);
  void set_challenges(Ctx& cx, const u64* betas, const u64* gammas, const u64* alphas);
  size_t ctx_bytes() const;
  void ensure_ctx(size_t count);
  void ensure_vals(int buf, size_t batch);

  MainStreamLease main_lease_;   // first member: given back also when the constructor throws further down
  Circuit c_;
  ProofLayout layout_;
  NttTables tables_;
  DeviceWitnessProgram wp_;
  DevMem cs_vals_, cs_coeffs_, cs_lde_, cs_tree_, k_is_, preamble_, l0_inv_;
  u64 digest_[4];
  std::vector<u64> cs_cap_;
  QuotientArgs qa_proto_;
  hipStream_t stream_ = nullptr;
  std::vector<std::unique_ptr<Ctx>> ctxs_;   // proofs in flight: one working set + HIP stream each
  hipEvent_t ev_witness_[2] = {nullptr, nullptr};  // witness pass into vals_[b] finished
  hipEvent_t ev_ext_ = nullptr, ev_main_ = nullptr;  // stream_join / wait_stream
  std::vector<hipEvent_t> marks_[MAX_MARKS];         // mark(slot): [0] the main stream, [1 + k] proving stream k
  size_t marks_recorded_[MAX_MARKS] = {0};           // how many of them the latest mark(slot) recorded
  int streams_ = 16;
  size_t pool_first_ = 0;      // this circuit's first position in the process-wide stream pool (prover.hip)
  bool single_proof_ = false;  // set per prove call: one proof in flight -> latency-oriented kernel forms
  DevMem vals_[2];             // witness values of a pass, slot-major [slot][proof of the pass]; double-buffered
  size_t vals_batch_[2] = {0, 0};
  size_t pass_counter_ = 0;    // witness passes issued so far (parity = buffer)
  std::vector<DevMem> owned_;
  bool kstats_on_ = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> kstats_pending_, kstats_free_;
  double kstats_ms_ = 0;
  u64 kstats_launches_ = 0;
};

}  // namespace p25
