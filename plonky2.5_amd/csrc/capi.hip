// C ABI of libp25 (declared in include/p25.h).  Thin: argument checks, device buffers, launches.
#include <stdlib.h>
#include <dirent.h>
#include <unistd.h>
#include <cstring>
#include <memory>
#include <atomic>
#include <mutex>
#include "../../include/p25.h"
#include "kernels.h"

namespace p25 {
thread_local std::string g_last_error;
// Process-wide device selection.  hipSetDevice is per host thread, so the chosen index is recorded and
// re-applied at every entry point: a worker thread of the host calling into the library lands on the GPU
// p25_device_init selected, not on device 0.  -1 = not initialised (first use picks the current device).
static std::atomic<int> g_device{-1};
static std::once_flag g_tables_once;
static std::unique_ptr<NttTables> g_tables;
static std::mutex g_primitives_mutex;  // the primitive entry points share g_tables (NttTables caches are not thread-safe)

// What p25_runtime_info reports about the hardware-queue setting (p25.h).  Filled by the first of p25_device_init[_ex] /
// ensure_device to run, before that call's first HIP call.
struct HwQueueState {
  std::mutex mu;
  bool decided = false;
  int requested = 0;          // what the library asked for (0 = nothing)
  bool host_exported = false; // GPU_MAX_HW_QUEUES was in the environment already: the host's value stands
  bool runtime_was_up = false;  // this process had the GPU driver open before the library's first HIP call
};
static HwQueueState g_hwq;

// True if this process already holds /dev/kfd: the ROCm runtime under HIP has been initialised (by the host, by torch, or by
// a profiler's preloaded tool library), so an environment variable set NOW may no longer be read.
static bool gpu_driver_open() {
  DIR* d = opendir("/proc/self/fd");
  if (!d) return false;
  bool found = false;
  char link[300], target[256];
  while (struct dirent* e = readdir(d)) {
    if (e->d_name[0] == '.') continue;
    snprintf(link, sizeof link, "/proc/self/fd/%s", e->d_name);
    const ssize_t k = readlink(link, target, sizeof target - 1);
    if (k <= 0) continue;
    target[k] = 0;
    if (!strcmp(target, "/dev/kfd")) {
      found = true;
      break;
    }
  }
  closedir(d);
  return found;
}

// Ask the HIP runtime for `hw_queues` hardware queues (never overriding a value the host exported) BEFORE the library's first
// HIP call, and remember whether that can still have had an effect.  Returns true when it probably had none.
static bool apply_hw_queues(int hw_queues) {
  std::lock_guard<std::mutex> l(g_hwq.mu);
  if (!g_hwq.decided) {
    g_hwq.decided = true;
    g_hwq.host_exported = getenv("GPU_MAX_HW_QUEUES") != nullptr;
    g_hwq.runtime_was_up = gpu_driver_open();
    g_hwq.requested = hw_queues;
    if (hw_queues > 0) setenv("GPU_MAX_HW_QUEUES", std::to_string(hw_queues).c_str(), 0);
  }
  return g_hwq.requested > 0 && !g_hwq.host_exported && g_hwq.runtime_was_up;
}
static const char* const HWQ_LATE_MSG =
    "warning: GPU_MAX_HW_QUEUES was not in the environment and the GPU runtime was already open in this process when libp25 "
    "first ran (the host, torch or a profiler touched HIP first): if HIP had been initialised too the setting has no effect "
    "and the 16 proving streams share the runtime's default 4 hardware queues (measured 78.8 vs 91.7+ proofs/s); export "
    "GPU_MAX_HW_QUEUES=24 before the process touches HIP";

NttTables& tables() {
  std::call_once(g_tables_once, [] { g_tables.reset(new NttTables()); });
  return *g_tables;
}

p25_status ensure_device() {
  int d = g_device.load(std::memory_order_acquire);
  if (d < 0) {
    // a host that skipped p25_device_init: the library's default still applies, before this first HIP call
    (void)apply_hw_queues(P25_DEFAULT_HW_QUEUES);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
      g_last_error = "no HIP device available (libp25 has no CPU fallback)";
      return P25_ERR_NO_DEVICE;
    }
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) cur = 0;
    int expected = -1;
    g_device.compare_exchange_strong(expected, cur, std::memory_order_acq_rel);
    d = g_device.load(std::memory_order_acquire);
  }
  if (hipSetDevice(d) != hipSuccess) {
    g_last_error = "hipSetDevice failed";
    return P25_ERR_HIP;
  }
  return P25_OK;
}

struct DevBuf {
  u64* p = nullptr;
  explicit DevBuf(size_t words) {
    if (words) P25_HIP(hipMalloc(&p, words * sizeof(u64)));
  }
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

template <class F>
p25_status guarded(F&& f) {
  try {
    p25_status s = ensure_device();
    if (s != P25_OK) return s;
    return f();
  } catch (const HipError& e) {
    g_last_error = e.what();
    return P25_ERR_HIP;
  } catch (const std::invalid_argument& e) {
    g_last_error = e.what();
    return P25_ERR_INVALID_ARG;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return P25_ERR_INTERNAL;
  }
}

static bool is_pow2(size_t x) { return x && !(x & (x - 1)); }

void lde_commit_dev(const u64* d_polys, unsigned log_n, size_t n_polys, bool from_coeffs,
                    unsigned rate_bits, unsigned cap_height, u64* d_coeffs, u64* d_tmp, u64* d_lde,
                    u64* d_tree, hipStream_t st) {
  const size_t n = (size_t)1 << log_n;
  const size_t big = n << rate_bits;
  if (!from_coeffs)     // values -> coefficients -> LDE: the prover's commit sequence (one combined entry, kernels.h)
    ntt_inverse_then_lde(tables(), d_polys, n, d_tmp, n, d_coeffs, n, d_lde, big, (int)log_n, (int)rate_bits, (int)n_polys,
                         gl::GENERATOR, st);
  else
    ntt_lde_bitrev(tables(), d_polys, n, d_lde, big, (int)log_n, (int)rate_bits, (int)n_polys, gl::GENERATOR, st);
  if (d_tree) launch_merkle_tree(d_lde, big, (int)n_polys, big, cap_height, d_tree, st);
}
}  // namespace p25

using namespace p25;

extern "C" {

const char* p25_last_error(void) { return g_last_error.c_str(); }
const char* p25_version(void) { return "libp25 0.1 (gfx950)"; }

// The prover keeps up to 16 proofs in flight on separate HIP streams so that the latency-bound stretches of
// one proof overlap the VALU-bound kernels of others.  ROCclr multiplexes streams onto
// GPU_MAX_HW_QUEUES hardware queues (default 4), and streams sharing a queue serialise: 16 queues
// measured 78.8 -> 91.7 proofs/s on one MI355X (12 streams); 24 queues with 16 streams a further 1.5%.  The variable is read
// when the HIP runtime initialises, so the library asks for it in p25_device_init_ex -- an explicit call with an explicit
// argument, before its first HIP call, never overriding a value the host has exported -- and NOT when it is loaded (rounds
// 1-4 did that from a constructor: a process-wide side effect a host had no say in).  A host that initialises HIP before
// that call exports the variable itself.
p25_status p25_device_init(int device_index) {
  const p25_status s = p25_device_init_ex(device_index, P25_DEFAULT_HW_QUEUES);
  return s == P25_WARN_HW_QUEUES_LATE ? P25_OK : s;   // the plain form never warns through its status (p25_runtime_info does)
}

p25_status p25_device_init_ex(int device_index, int hw_queues) {
  if (hw_queues < 0 || hw_queues > 128) {
    g_last_error = "hw_queues out of range (0 = leave the runtime's setting alone)";
    return P25_ERR_INVALID_ARG;
  }
  const bool late = apply_hw_queues(hw_queues);   // before the first HIP call below
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    g_last_error = "no HIP device available (libp25 has no CPU fallback)";
    return P25_ERR_NO_DEVICE;
  }
  if (device_index < 0 || device_index >= n) {
    g_last_error = "device index out of range";
    return P25_ERR_INVALID_ARG;
  }
  if (hipSetDevice(device_index) != hipSuccess) {
    g_last_error = "hipSetDevice failed";
    return P25_ERR_HIP;
  }
  g_device.store(device_index, std::memory_order_release);
  if (late) {
    g_last_error = HWQ_LATE_MSG;
    return P25_WARN_HW_QUEUES_LATE;
  }
  return P25_OK;
}

p25_status p25_runtime_info(p25_runtime_info_t* out) {
  if (!out) {
    g_last_error = "out is null";
    return P25_ERR_INVALID_ARG;
  }
  memset(out, 0, sizeof *out);
  out->device_index = g_device.load(std::memory_order_acquire);
  {
    std::lock_guard<std::mutex> l(g_hwq.mu);
    out->hw_queues_requested = g_hwq.requested;
    out->hw_queues_host_exported = g_hwq.host_exported ? 1 : 0;
    out->runtime_open_before_init = g_hwq.runtime_was_up ? 1 : 0;
    out->hw_queues_setting_late = (g_hwq.decided && g_hwq.requested > 0 && !g_hwq.host_exported && g_hwq.runtime_was_up) ? 1 : 0;
  }
  const char* e = getenv("GPU_MAX_HW_QUEUES");
  out->hw_queues_env = e ? (int32_t)strtol(e, nullptr, 10) : 0;
  out->proving_streams = 16;
  out->main_streams = 2;
  return P25_OK;
}

p25_status p25_shader_clock_hz(double* hz_out) {
  return guarded([&]() -> p25_status {
    if (!hz_out) throw std::invalid_argument("hz_out is null");
    *hz_out = measure_shader_clock_hz(0);
    return P25_OK;
  });
}

p25_status p25_poseidon_permute(uint64_t* states, size_t n) {
  return guarded([&]() -> p25_status {
    if (!states && n) throw std::invalid_argument("states is null");
    DevBuf d(n * 12);
    P25_HIP(hipMemcpy(d.p, states, n * 96, hipMemcpyHostToDevice));
    launch_poseidon_permute(d.p, n, 0);
    P25_HIP(hipGetLastError());
    P25_HIP(hipMemcpy(states, d.p, n * 96, hipMemcpyDeviceToHost));
    return P25_OK;
  });
}

p25_status p25_poseidon2_permute(uint64_t* states, size_t n) {
  return guarded([&]() -> p25_status {
    if (!states && n) throw std::invalid_argument("states is null");
    DevBuf d(n * 12);
    P25_HIP(hipMemcpy(d.p, states, n * 96, hipMemcpyHostToDevice));
    launch_poseidon2_permute(d.p, n, 0);
    P25_HIP(hipGetLastError());
    P25_HIP(hipMemcpy(states, d.p, n * 96, hipMemcpyDeviceToHost));
    return P25_OK;
  });
}

size_t p25_merkle_tree_words(size_t n_leaves, unsigned cap_height) {
  if (!is_pow2(n_leaves) || cap_height > 63 || n_leaves < ((size_t)1 << cap_height)) return 0;
  return merkle_tree_words(n_leaves, cap_height);
}

p25_status p25_merkle_commit(const uint64_t* leaves_cm, size_t n_leaves, size_t width,
                             unsigned cap_height, uint64_t* cap_out, uint64_t* tree_out) {
  return guarded([&]() -> p25_status {
    if (!is_pow2(n_leaves) || cap_height > 40 || n_leaves < ((size_t)1 << cap_height) || !width ||
        width > (1u << 20) || !leaves_cm)
      throw std::invalid_argument("p25_merkle_commit: bad shape");
    DevBuf d(n_leaves * width);
    size_t tw = merkle_tree_words(n_leaves, cap_height);
    DevBuf t(tw);
    P25_HIP(hipMemcpy(d.p, leaves_cm, n_leaves * width * 8, hipMemcpyHostToDevice));
    u64* cap = launch_merkle_tree(d.p, n_leaves, (int)width, n_leaves, cap_height, t.p, 0);
    P25_HIP(hipGetLastError());
    if (cap_out) P25_HIP(hipMemcpy(cap_out, cap, ((size_t)32) << cap_height, hipMemcpyDeviceToHost));
    if (tree_out) P25_HIP(hipMemcpy(tree_out, t.p, tw * 8, hipMemcpyDeviceToHost));
    P25_HIP(hipDeviceSynchronize());
    return P25_OK;
  });
}

p25_status p25_merkle_commit_dev(const uint64_t* d_leaves_cm, size_t col_stride, size_t n_leaves,
                                 size_t width, unsigned cap_height, uint64_t* d_tree, void* stream) {
  return guarded([&]() -> p25_status {
    if (!is_pow2(n_leaves) || cap_height > 40 || n_leaves < ((size_t)1 << cap_height) || !width ||
        col_stride < n_leaves || !d_leaves_cm || !d_tree)
      throw std::invalid_argument("p25_merkle_commit_dev: bad shape");
    launch_merkle_tree(d_leaves_cm, col_stride, (int)width, n_leaves, cap_height, d_tree,
                       (hipStream_t)stream);
    P25_HIP(hipGetLastError());
    return P25_OK;
  });
}

p25_status p25_poseidon_permute_dev(uint64_t* d_states, size_t n, void* stream) {
  return guarded([&]() -> p25_status {
    launch_poseidon_permute(d_states, n, (hipStream_t)stream);
    P25_HIP(hipGetLastError());
    return P25_OK;
  });
}

p25_status p25_lde_commit(const uint64_t* polys, unsigned log_n, size_t n_polys, int from_coeffs,
                          unsigned rate_bits, unsigned cap_height, uint64_t* coeffs_out,
                          uint64_t* lde_out, uint64_t* cap_out) {
  return guarded([&]() -> p25_status {
    if (!polys || !n_polys || log_n > 20 || rate_bits > 3 || log_n + rate_bits < cap_height)
      throw std::invalid_argument("p25_lde_commit: bad shape");
    std::lock_guard<std::mutex> lk(g_primitives_mutex);
    const size_t n = (size_t)1 << log_n, big = n << rate_bits;
    DevBuf in(n * n_polys), co(n * n_polys), tmp(n * n_polys), lde(big * n_polys);
    size_t tw = merkle_tree_words(big, cap_height);
    DevBuf tree(tw);
    P25_HIP(hipMemcpy(in.p, polys, n * n_polys * 8, hipMemcpyHostToDevice));
    lde_commit_dev(in.p, log_n, n_polys, from_coeffs != 0, rate_bits, cap_height, co.p, tmp.p, lde.p,
                   tree.p, 0);
    P25_HIP(hipGetLastError());
    P25_HIP(hipDeviceSynchronize());
    if (coeffs_out)
      P25_HIP(hipMemcpy(coeffs_out, from_coeffs ? in.p : co.p, n * n_polys * 8, hipMemcpyDeviceToHost));
    if (lde_out) P25_HIP(hipMemcpy(lde_out, lde.p, big * n_polys * 8, hipMemcpyDeviceToHost));
    if (cap_out)
      P25_HIP(hipMemcpy(cap_out, tree.p + tw - ((size_t)4 << cap_height), ((size_t)32) << cap_height,
                        hipMemcpyDeviceToHost));
    return P25_OK;
  });
}

p25_status p25_lde_commit_dev(const uint64_t* d_polys, unsigned log_n, size_t n_polys, int from_coeffs,
                              unsigned rate_bits, unsigned cap_height, uint64_t* d_coeffs,
                              uint64_t* d_tmp, uint64_t* d_lde, uint64_t* d_tree, void* stream) {
  return guarded([&]() -> p25_status {
    if (!d_polys || !n_polys || log_n > 20 || rate_bits > 3 || !d_lde ||
        (!from_coeffs && (!d_coeffs || !d_tmp)))
      throw std::invalid_argument("p25_lde_commit_dev: bad shape");
    std::lock_guard<std::mutex> lk(g_primitives_mutex);
    lde_commit_dev(d_polys, log_n, n_polys, from_coeffs != 0, rate_bits, cap_height, d_coeffs, d_tmp,
                   d_lde, d_tree, (hipStream_t)stream);
    P25_HIP(hipGetLastError());
    return P25_OK;
  });
}

}  // extern "C"

// ======================================================================================
// circuits, proving, data formats
// ======================================================================================
#include "circuit_io.h"
#include "circuit_bytes.h"
#include "json_io.h"
#include "p3_circuit.h"
#include "p3_prover.h"
#include "prover.h"
#include "recursion.h"

struct p25_circuit {
  p25::Circuit circuit;                       // host tables (moved into dev on first device use)
  std::unique_ptr<p25::DeviceCircuit> dev;
  std::unique_ptr<p25::WitnessProgram> wp_info;
  bool moved = false;
  int streams = 16;  // proofs in flight (p25_circuit_set_streams)
  // Entry points that touch the device state of ONE circuit are serialised: upstream's `prove(&self)` is re-entrant,
  // so a host with a thread pool may call into the same circuit concurrently; here those calls queue up instead of
  // racing for the circuit's streams and contexts.  Different circuits never contend.
  mutable std::recursive_mutex mu;   // mutable: the read-only entry points (const handles) lock too
  const p25::Circuit& c() const { return dev ? dev->circuit() : circuit; }
  p25::DeviceCircuit& device() {
    if (!dev) {
      dev.reset(new p25::DeviceCircuit(std::move(circuit)));
      dev->set_streams(streams);
      moved = true;
    }
    return *dev;
  }
};
#define P25_LOCK(c) std::lock_guard<std::recursive_mutex> p25_lock_((c)->mu)

template <class F>
static p25_status host_guarded(F&& f) {
  try {
    return f();
  } catch (const p25::HipError& e) {
    p25::g_last_error = e.what();
    return P25_ERR_HIP;
  } catch (const std::invalid_argument& e) {
    p25::g_last_error = e.what();
    return P25_ERR_INVALID_ARG;
  } catch (const std::exception& e) {
    p25::g_last_error = e.what();
    return P25_ERR_INTERNAL;
  }
}

extern "C" {

static p25::P3Config checked_p3_config(const p25_p3_config* cfg) {
  // one quotient chunk is the reference's proof model (proof.rs:41-48); two (constraint degree 3) is the round-5 extension,
  // four / eight (degree 4-5 / 6-9, with log_blowup 2 / 3) round 6's
  if (cfg->log_quotient_degree < 0 || cfg->log_quotient_degree > 3)
    throw std::invalid_argument("1, 2, 4 or 8 quotient chunks are supported (log_quotient_degree 0..3)");
  if (cfg->log_quotient_degree > cfg->log_blowup) throw std::invalid_argument("log_quotient_degree above log_blowup");
  if (cfg->opening_matrix_log_max_height != cfg->log_trace_height + cfg->log_blowup)
    throw std::invalid_argument("opening_matrix_log_max_height must be log_trace_height + log_blowup");
  if (cfg->trace_width < 1 || cfg->trace_width > 64 || cfg->log_trace_height < 1 || cfg->log_trace_height > 24 ||
      cfg->num_queries < 1 || cfg->num_queries > 1000 || cfg->degree_bits < 1 || cfg->degree_bits > cfg->log_trace_height ||
      cfg->opening_matrix_log_max_height < 1 || cfg->opening_matrix_log_max_height > 30 || cfg->quotient_opened_len < 1 ||
      cfg->log_blowup < 1 || cfg->log_blowup > 4 || cfg->proof_of_work_bits < 0 || cfg->proof_of_work_bits > 32)
    throw std::invalid_argument("p25_p3_config out of range");
  p25::P3Config pc;
  pc.fri_config.log_blowup = cfg->log_blowup;
  pc.fri_config.num_queries = cfg->num_queries;
  pc.fri_config.proof_of_work_bits = cfg->proof_of_work_bits;
  pc.log_quotient_degree = cfg->log_quotient_degree;
  pc.log_trace_height = cfg->log_trace_height;
  pc.trace_width = cfg->trace_width;
  pc.opening_matrix_log_max_height = cfg->opening_matrix_log_max_height;
  pc.opening_proof_query_openings_opened_values_length = cfg->quotient_opened_len;
  pc.degree_bits = cfg->degree_bits;
  return pc;
}
static p25::AirProgram air_from_c(const p25_air* air) {
  if (!air || !air->nodes || !air->constraints) throw std::invalid_argument("null AIR");
  if (air->n_nodes > (1u << 20) || air->n_constraints > (1u << 16)) throw std::invalid_argument("AIR too large");
  p25::AirProgram p;
  p.width = (int)air->width;
  for (uint32_t i = 0; i < air->n_nodes; i++)
    p.nodes.push_back(p25::AirProgram::Node{air->nodes[i].op, air->nodes[i].a, air->nodes[i].b, air->nodes[i].value});
  for (uint32_t i = 0; i < air->n_constraints; i++)
    p.constraints.push_back(p25::AirProgram::Constraint{air->constraints[i].node, air->constraints[i].when});
  p.validate();
  return p;
}
static p25_status build_verifier(const p25_p3_config* cfg, const p25::Air& air, p25_circuit** out) {
  p25::P3Config pc = checked_p3_config(cfg);
  if (pc.trace_width != air.width()) throw std::invalid_argument("Invalid Proof Shape");
  p25::CircuitBuilder cb;
  p25::p3_verify_proof(cb, pc, air);
  auto* h = new p25_circuit();
  h->circuit = cb.build();
  *out = h;
  return P25_OK;
}

p25_status p25_circuit_build_p3_verifier(const p25_p3_config* cfg, int32_t air, p25_circuit** out) {
  return host_guarded([&]() -> p25_status {
    if (!cfg || !out) throw std::invalid_argument("null argument");
    if (air != P25_AIR_FIBONACCI) throw std::invalid_argument("unknown AIR");
    p25::FibonacciAir fib;
    return build_verifier(cfg, fib, out);
  });
}
p25_status p25_circuit_build_p3_verifier_air(const p25_p3_config* cfg, const p25_air* air, p25_circuit** out) {
  return host_guarded([&]() -> p25_status {
    if (!cfg || !air || !out) throw std::invalid_argument("null argument");
    p25::AirProgram prog = air_from_c(air);
    // the chunk count follows from the AIR (uni-stark get_log_quotient_degree): the shape must say the same
    if (prog.log_quotient_degree() != cfg->log_quotient_degree)
      throw std::invalid_argument("an AIR of constraint degree " + std::to_string(prog.max_constraint_degree()) + " has 2^" +
                                  std::to_string(prog.log_quotient_degree()) + " quotient chunks: log_quotient_degree must say so");
    p25::ProgramAir pa(std::move(prog));
    return build_verifier(cfg, pa, out);
  });
}

p25_status p25_circuit_build_gadget(int32_t kind, int32_t param, p25_circuit** out) {
  return host_guarded([&]() -> p25_status {
    if (!out) throw std::invalid_argument("null argument");
    auto* h = new p25_circuit();
    try {
      h->circuit = p25::build_gadget_circuit(kind, param);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return P25_OK;
  });
}

p25_status p25_circuit_build_gate_eval(int32_t kind, p25_circuit** out) {
  return host_guarded([&]() -> p25_status {
    if (!out) throw std::invalid_argument("null argument");
    if (kind < 0 || kind >= p25::G_NUM_KINDS) throw std::invalid_argument("unknown gate kind");
    auto* h = new p25_circuit();
    try {
      h->circuit = p25::build_gate_eval_circuit((p25::GateKind)kind);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return P25_OK;
  });
}
static p25_status build_recursive(p25_circuit* inner, const uint64_t* digest4, const uint64_t* cs_cap, int32_t n_proofs,
                                  bool expose_commitment, p25_circuit** out) {
  auto body = [&]() -> p25_status {
    if (!inner || !out) throw std::invalid_argument("null argument");
    if ((digest4 == nullptr) != (cs_cap == nullptr)) throw std::invalid_argument("pass both digest4 and cs_cap, or neither");
    P25_LOCK(inner);   // reads inner->c(): another thread's first device use moves the host tables
    const p25::Circuit& ic = inner->c();
    uint64_t dg[4];
    std::vector<u64> cap((size_t)4 << ic.cfg.cap_height);
    if (digest4) {
      memcpy(dg, digest4, 32);
      memcpy(cap.data(), cs_cap, cap.size() * 8);
      for (u64 v : cap)
        if (v >= gl::P) throw std::invalid_argument("non-canonical cap word");
      for (u64 v : dg)
        if (v >= gl::P) throw std::invalid_argument("non-canonical digest word");
    } else {
      p25::DeviceCircuit& d = inner->device();  // needs the GPU: commits the constants/sigmas polynomials
      memcpy(dg, d.digest(), 32);
      cap = d.cs_cap();
    }
    auto* h = new p25_circuit();
    try {
      h->circuit = p25::build_recursive_verifier(inner->c(), dg, cap, n_proofs, expose_commitment);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return P25_OK;
  };
  return digest4 ? host_guarded(body) : guarded(body);
}
p25_status p25_circuit_build_recursive_verifier(p25_circuit* inner, const uint64_t* digest4, const uint64_t* cs_cap,
                                                int32_t n_proofs, p25_circuit** out) {
  return build_recursive(inner, digest4, cs_cap, n_proofs, false, out);
}
p25_status p25_circuit_build_aggregator(p25_circuit* inner, const uint64_t* digest4, const uint64_t* cs_cap,
                                        int32_t n_proofs, p25_circuit** out) {
  return build_recursive(inner, digest4, cs_cap, n_proofs, true, out);
}

p25_status p25_circuit_export(const p25_circuit* c, uint8_t* buf, size_t cap, size_t* len_out) {
  return host_guarded([&]() -> p25_status {
    if (!c || !len_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    std::vector<uint8_t> b = p25::circuit_to_blob(c->c());
    *len_out = b.size();
    if (buf) {
      if (cap < b.size()) throw std::invalid_argument("buffer too small");
      memcpy(buf, b.data(), b.size());
    }
    return P25_OK;
  });
}
p25_status p25_circuit_to_bytes(p25_circuit* c, uint8_t** bytes_out, size_t* len_out) {
  return guarded([&]() -> p25_status {
    if (!c || !bytes_out || !len_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    p25::DeviceCircuit& d = c->device();   // the constants/sigmas commitment is computed on the GPU
    std::vector<u64> coeffs, lde, tree;
    d.commitment_to_host(coeffs, lde, tree);
    p25::CircuitCommitment cm;
    cm.coeffs = coeffs.data();
    cm.lde = lde.data();
    cm.tree = tree.data();
    memcpy(cm.digest, d.digest(), 32);
    cm.public_inputs = d.circuit().public_inputs;
    std::vector<uint8_t> b = p25::circuit_data_to_bytes(d.circuit(), cm);
    uint8_t* m = (uint8_t*)malloc(b.size() ? b.size() : 1);
    if (!m) throw std::runtime_error("out of host memory");
    memcpy(m, b.data(), b.size());
    *bytes_out = m;
    *len_out = b.size();
    return P25_OK;
  });
}
void p25_free(void* p) { free(p); }
p25_status p25_circuit_from_bytes(const uint8_t* bytes, size_t len, const uint32_t* input_targets, size_t n_inputs,
                                  uint64_t* digest4_out, p25_circuit** out) {
  return host_guarded([&]() -> p25_status {
    if (!bytes || !out || (!input_targets && n_inputs)) throw std::invalid_argument("null argument");
    auto* h = new p25_circuit();
    try {
      u64 dg[4];
      h->circuit = p25::circuit_data_from_bytes(bytes, len, input_targets, n_inputs, dg);
      if (digest4_out) memcpy(digest4_out, dg, 32);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return P25_OK;
  });
}
p25_status p25_circuit_input_targets(const p25_circuit* c, uint32_t* targets_out, size_t cap, size_t* n_out) {
  return host_guarded([&]() -> p25_status {
    if (!c || !n_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    const p25::Circuit& k = c->c();
    *n_out = k.input_targets.size();
    if (targets_out) {
      if (cap < k.input_targets.size()) throw std::invalid_argument("buffer too small");
      for (size_t i = 0; i < k.input_targets.size(); i++) targets_out[i] = (uint32_t)k.target_index(k.input_targets[i]);
    }
    return P25_OK;
  });
}
p25_status p25_circuit_import(const uint8_t* blob, size_t len, p25_circuit** out) {
  return host_guarded([&]() -> p25_status {
    if (!blob || !out) throw std::invalid_argument("null argument");
    auto* h = new p25_circuit();
    try {
      h->circuit = p25::circuit_from_blob(blob, len);
    } catch (...) {
      delete h;
      throw;
    }
    *out = h;
    return P25_OK;
  });
}
void p25_circuit_destroy(p25_circuit* c) { delete c; }

p25_status p25_circuit_info(p25_circuit* c, p25_circuit_info_t* out) {
  return host_guarded([&]() -> p25_status {
    if (!c || !out) throw std::invalid_argument("null argument");
    P25_LOCK(c);   // reads c->c() and builds wp_info lazily: serialised with a concurrent first prove()
    const p25::Circuit& k = c->c();
    memset(out, 0, sizeof(*out));
    out->degree_bits = k.degree_bits;
    size_t used = 0;
    for (auto& r : k.rows) used += r.kind != p25::G_NOOP;
    out->num_rows_used = used;
    out->num_wires = k.cfg.num_wires;
    out->num_routed_wires = k.cfg.num_routed_wires;
    out->num_inputs = k.input_targets.size();
    out->num_generators = k.generators.size();
    out->num_gate_types = k.gates.size();
    out->num_selectors = k.num_selectors;
    out->num_constants_sigmas = k.constants_sigmas.size();
    out->num_gate_constraints = k.num_gate_constraints;
    out->proof_words = p25::make_proof_layout(k).total;
    if (!c->wp_info) c->wp_info.reset(new p25::WitnessProgram(p25::build_witness_program(k)));
    out->witness_levels = c->wp_info->level_start.size() - 1;
    out->witness_slots = c->wp_info->num_slots;
    out->num_random_fill = c->wp_info->num_random_fill;
    out->num_public_inputs = k.public_inputs.size();
    out->num_challenges = k.cfg.num_challenges;
    out->num_partial_products = k.num_partial_products;
    out->quotient_degree_factor = k.cfg.max_quotient_degree_factor;
    return P25_OK;
  });
}
p25_status p25_circuit_gate_counts(const p25_circuit* c, uint64_t* counts_out, size_t cap, char* ids_out, size_t ids_cap) {
  return host_guarded([&]() -> p25_status {
    if (!c || !counts_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    const p25::Circuit& k = c->c();
    if (cap < k.gates.size()) throw std::invalid_argument("buffer too small");
    std::string ids;
    for (size_t i = 0; i < k.gates.size(); i++) {
      size_t n = 0;
      for (auto& r : k.rows) n += r.kind == k.gates[i];
      counts_out[i] = n;
      ids += p25::gate_info(k.gates[i]).id;
      ids += '\n';
    }
    if (ids_out && ids_cap) snprintf(ids_out, ids_cap, "%s", ids.c_str());
    return P25_OK;
  });
}
p25_status p25_circuit_digest(p25_circuit* c, uint64_t* digest4, uint64_t* cs_cap) {
  return guarded([&]() -> p25_status {
    if (!c || !digest4) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    p25::DeviceCircuit& d = c->device();
    memcpy(digest4, d.digest(), 32);
    if (cs_cap) memcpy(cs_cap, d.cs_cap().data(), d.cs_cap().size() * 8);
    return P25_OK;
  });
}

static void fill_timings(p25_timings* t, const p25::PhaseTimes& pt) {
  if (!t) return;
  t->witness_ms = pt.witness; t->wires_commit_ms = pt.wires_commit; t->partial_products_ms = pt.zs_pp;
  t->zs_commit_ms = pt.zs_commit; t->quotient_ms = pt.quotient; t->quotient_commit_ms = pt.quotient_commit;
  t->openings_ms = pt.openings; t->fri_ms = pt.fri; t->total_ms = pt.total;
}

p25_status p25_prove_batch(p25_circuit* c, const uint64_t* inputs, size_t n_proofs, const uint64_t* seeds,
                           uint64_t* proofs_out, size_t proof_stride_words, p25_status* per_proof_status,
                           p25_timings* timings) {
  return guarded([&]() -> p25_status {
    if (!c || !inputs || !proofs_out || !per_proof_status) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    if (!n_proofs) return P25_OK;
    p25::PhaseTimes pt;
    c->device().prove_batch(inputs, n_proofs, seeds, proofs_out, proof_stride_words, per_proof_status, timings ? &pt : nullptr);
    fill_timings(timings, pt);
    return P25_OK;
  });
}
p25_status p25_prove_batch_filler(p25_circuit* c, const uint64_t* inputs, size_t n_proofs, const uint64_t* filler,
                                  uint64_t* proofs_out, size_t proof_stride_words, p25_status* per_proof_status) {
  return guarded([&]() -> p25_status {
    if (!c || !inputs || !filler || !proofs_out || !per_proof_status) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    if (!n_proofs) return P25_OK;
    c->device().prove_batch(inputs, n_proofs, nullptr, proofs_out, proof_stride_words, per_proof_status, nullptr, filler);
    return P25_OK;
  });
}
p25_status p25_prove_batch_dev(p25_circuit* c, const uint64_t* d_inputs, size_t n_proofs, const uint64_t* d_seeds,
                               uint64_t* d_proofs, size_t proof_stride_words, uint32_t* d_status, p25_timings* timings) {
  return guarded([&]() -> p25_status {
    if (!c || !d_inputs || !d_seeds || !d_proofs || !d_status) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    p25::DeviceCircuit& d = c->device();
    if (proof_stride_words < d.layout().total) throw std::invalid_argument("proof_stride smaller than the proof");
    p25::PhaseTimes pt;
    d.prove_batch_dev(d_inputs, n_proofs, d_seeds, d_proofs, proof_stride_words, d_status, timings ? &pt : nullptr);
    fill_timings(timings, pt);
    return P25_OK;
  });
}
p25_status p25_prove_batch_dev_windows(p25_circuit* c, const uint64_t* d_buffer, size_t window_stride_words,
                                       size_t last_window_offset_words, size_t n_proofs, const uint64_t* d_seeds,
                                       uint64_t* d_proofs, size_t proof_stride_words, uint32_t* d_status) {
  return guarded([&]() -> p25_status {
    if (!c || !d_buffer || !d_seeds || !d_proofs || !d_status) throw std::invalid_argument("null argument");
    if (window_stride_words == 0) throw std::invalid_argument("window stride is zero");
    // every offset the witness pass forms, (first proof + p) * stride, stays below 2^60 words: no size_t product can wrap
    if (n_proofs > ((size_t)1 << 32) || window_stride_words > ((size_t)1 << 60) / (n_proofs ? n_proofs : 1))
      throw std::invalid_argument("windows out of range (n_proofs * window_stride_words must stay below 2^60)");
    if (n_proofs && last_window_offset_words > (n_proofs - 1) * window_stride_words)
      throw std::invalid_argument("last window lies beyond the windows before it");
    P25_LOCK(c);
    p25::DeviceCircuit& d = c->device();
    if (proof_stride_words < d.layout().total) throw std::invalid_argument("proof_stride smaller than the proof");
    d.prove_batch_dev(d_buffer, n_proofs, d_seeds, d_proofs, proof_stride_words, d_status, nullptr, nullptr,
                      window_stride_words, last_window_offset_words);
    return P25_OK;
  });
}
p25_status p25_circuit_set_streams(p25_circuit* c, int32_t n_streams) {
  return host_guarded([&]() -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    if (n_streams < 1 || n_streams > 32) throw std::invalid_argument("n_streams must be in 1..32");
    P25_LOCK(c);
    c->streams = n_streams;
    if (c->dev) c->dev->set_streams(n_streams);
    return P25_OK;
  });
}
p25_status p25_circuit_sync(p25_circuit* c) {
  return guarded([&]() -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    c->device().sync();
    return P25_OK;
  });
}
p25_status p25_circuit_stream_join(p25_circuit* c, void* stream) {
  return guarded([&]() -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    c->device().stream_join((hipStream_t)stream);
    return P25_OK;
  });
}
p25_status p25_circuit_wait_stream(p25_circuit* c, void* stream) {
  return guarded([&]() -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    c->device().wait_stream((hipStream_t)stream);
    return P25_OK;
  });
}
p25_status p25_circuit_mark(p25_circuit* c, uint32_t slot) {
  return guarded([&]() -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    c->device().mark((int)slot);
    return P25_OK;
  });
}
p25_status p25_circuit_stream_wait_mark(p25_circuit* c, uint32_t slot, void* stream) {
  return guarded([&]() -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    c->device().stream_wait_mark((hipStream_t)stream, (int)slot);
    return P25_OK;
  });
}
p25_status p25_circuit_wait_mark(p25_circuit* c, p25_circuit* producer, uint32_t slot) {
  return guarded([&]() -> p25_status {
    if (!c || !producer) throw std::invalid_argument("null argument");
    if (c == producer) throw std::invalid_argument("a circuit's own streams are already in order");
    std::lock(c->mu, producer->mu);   // both, without a lock-order deadlock between two threads chaining opposite ways
    std::lock_guard<std::recursive_mutex> l1(c->mu, std::adopt_lock), l2(producer->mu, std::adopt_lock);
    c->device().wait_mark(producer->device(), (int)slot);
    return P25_OK;
  });
}
p25_status p25_circuit_kernel_stats(p25_circuit* c, int enable, int reset, double* ms_out, uint64_t* launches_out) {
  return guarded([&]() -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    p25::DeviceCircuit& d = c->device();
    d.kernel_stats_enable(enable != 0);
    u64 n = 0;
    d.kernel_stats(ms_out, &n, reset != 0);
    if (launches_out) *launches_out = n;
    return P25_OK;
  });
}
p25_status p25_witness(p25_circuit* c, const uint64_t* inputs, uint64_t seed, uint64_t* wires_out, p25_status* proof_status) {
  return guarded([&]() -> p25_status {
    if (!c || !inputs || !wires_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    int32_t st = c->device().witness(inputs, seed, wires_out);
    if (proof_status) *proof_status = st;
    return P25_OK;
  });
}

p25_status p25_transcript(const uint64_t* observe, const uint32_t* seg_len, const uint32_t* n_challenges,
                          size_t n_segments, uint64_t* challenges_out) {
  return guarded([&]() -> p25_status {
    if ((n_segments && (!seg_len || !n_challenges)) || !challenges_out) throw std::invalid_argument("null argument");
    size_t n_obs = 0;
    for (size_t k = 0; k < n_segments; k++) n_obs += seg_len[k];
    if (n_obs && !observe) throw std::invalid_argument("null argument");
    p25::transcript_script(observe, seg_len, n_challenges, n_segments, challenges_out);
    return P25_OK;
  });
}
p25_status p25_partial_products(p25_circuit* c, const uint64_t* wires, const uint64_t* betas, const uint64_t* gammas,
                                uint64_t* out) {
  return guarded([&]() -> p25_status {
    if (!c || !wires || !betas || !gammas || !out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    c->device().partial_products(wires, betas, gammas, out);
    return P25_OK;
  });
}
p25_status p25_quotient(p25_circuit* c, const uint64_t* wires, const uint64_t* zs_pp, const uint64_t* betas,
                        const uint64_t* gammas, const uint64_t* alphas, uint64_t* out) {
  return guarded([&]() -> p25_status {
    if (!c || !wires || !zs_pp || !betas || !gammas || !alphas || !out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    c->device().quotient(wires, zs_pp, betas, gammas, alphas, out);
    return P25_OK;
  });
}
p25_status p25_eval_polys(const uint64_t* coeffs, size_t n_polys, unsigned log_n, const uint64_t* point, uint64_t scale,
                          uint64_t* out) {
  return guarded([&]() -> p25_status {
    if (!coeffs || !point || !out) throw std::invalid_argument("null argument");
    if (!n_polys || n_polys > (1u << 16) || log_n > 22) throw std::invalid_argument("p25_eval_polys: bad shape");
    if (point[0] >= gl::P || point[1] >= gl::P || scale >= gl::P) throw std::invalid_argument("non-canonical point");
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n_polys * n; i++)
      if (coeffs[i] >= gl::P) throw std::invalid_argument("non-canonical coefficient");
    const size_t chunks = log_n > 16 ? ((size_t)1 << (log_n - 16)) : 1;
    DevBuf d_c(n_polys * n), d_pt(2), d_scr(2 * 1026 + 2 * n_polys * chunks), d_out(2 * n_polys);
    P25_HIP(hipMemcpy(d_c.p, coeffs, n_polys * n * 8, hipMemcpyHostToDevice));
    P25_HIP(hipMemcpy(d_pt.p, point, 16, hipMemcpyHostToDevice));
    p25::launch_eval_polys(d_c.p, (uint32_t)n_polys, log_n, d_pt.p, scale, d_scr.p, d_out.p, 0);
    P25_HIP(hipGetLastError());
    P25_HIP(hipMemcpy(out, d_out.p, 2 * n_polys * 8, hipMemcpyDeviceToHost));
    return P25_OK;
  });
}
static bool fri_shape_from_c(unsigned log_n, unsigned rate_bits, unsigned cap_height, const int32_t* arity_bits,
                             size_t n_layers, unsigned pow_bits, unsigned num_queries, p25::FriShape& sh) {
  if (log_n < 1 || log_n > 22 || rate_bits > 3 || cap_height > 16 || n_layers > 8 || (n_layers && !arity_bits) ||
      num_queries < 1 || num_queries > 64 || pow_bits > 32)
    return false;
  sh.log_n = (int)log_n;
  sh.rate_bits = (int)rate_bits;
  sh.cap_height = cap_height;
  sh.arity_bits.assign(arity_bits, arity_bits + n_layers);
  sh.pow_bits = (int)pow_bits;
  sh.num_queries = (int)num_queries;
  int deg = (int)log_n, bits = (int)(log_n + rate_bits);
  if (bits < (int)cap_height) return false;
  for (int a : sh.arity_bits) {
    if (a < 1 || a > 8) return false;
    deg -= a;
    bits -= a;
    if (deg < 0 || bits < (int)cap_height) return false;
  }
  return true;
}
size_t p25_fri_prove_words(unsigned log_n, unsigned rate_bits, unsigned cap_height, const int32_t* arity_bits,
                           size_t n_layers, unsigned num_queries) {
  p25::FriShape sh;
  if (!fri_shape_from_c(log_n, rate_bits, cap_height, arity_bits, n_layers, 0, num_queries, sh)) return 0;
  return p25::fri_prove_words(sh);
}
p25_status p25_fri_prove(const uint64_t* coeffs, unsigned log_n, unsigned rate_bits, unsigned cap_height,
                         const int32_t* arity_bits, size_t n_layers, unsigned pow_bits, unsigned num_queries,
                         const uint64_t* seed, size_t n_seed, uint64_t* out, size_t out_cap, p25_status* status_out) {
  return guarded([&]() -> p25_status {
    if (!coeffs || !out || !status_out || (n_seed && !seed)) throw std::invalid_argument("null argument");
    p25::FriShape sh;
    if (!fri_shape_from_c(log_n, rate_bits, cap_height, arity_bits, n_layers, pow_bits, num_queries, sh))
      throw std::invalid_argument("p25_fri_prove: bad shape");
    if (out_cap < p25::fri_prove_words(sh)) throw std::invalid_argument("buffer too small");
    std::lock_guard<std::mutex> lk(g_primitives_mutex);
    int32_t st = 0;
    p25::fri_prove_standalone(tables(), coeffs, sh, seed, n_seed, out, &st);
    *status_out = st;
    return P25_OK;
  });
}

p25_status p25_p3_proof_from_json(const char* json, size_t len, uint64_t* inputs_out, size_t cap, size_t* n_out,
                                  p25_p3_config* cfg_out) {
  p25_status s = host_guarded([&]() -> p25_status {
    if (!json || !n_out) throw std::invalid_argument("null argument");
    std::vector<u64> in;
    p25::P3Config pc;
    p25::p3_proof_from_json(json, len, in, pc);
    *n_out = in.size();
    if (inputs_out) {
      if (cap < in.size()) throw std::invalid_argument("buffer too small");
      memcpy(inputs_out, in.data(), in.size() * 8);
    }
    if (cfg_out) {
      cfg_out->log_blowup = pc.fri_config.log_blowup;
      cfg_out->num_queries = pc.fri_config.num_queries;
      cfg_out->proof_of_work_bits = pc.fri_config.proof_of_work_bits;
      cfg_out->log_quotient_degree = pc.log_quotient_degree;
      cfg_out->log_trace_height = pc.log_trace_height;
      cfg_out->trace_width = pc.trace_width;
      cfg_out->opening_matrix_log_max_height = pc.opening_matrix_log_max_height;
      cfg_out->quotient_opened_len = pc.opening_proof_query_openings_opened_values_length;
      cfg_out->degree_bits = pc.degree_bits;
    }
    return P25_OK;
  });
  if (s == P25_ERR_INVALID_ARG && p25::g_last_error.rfind("p3 proof JSON", 0) == 0) return P25_ERR_PARSE;
  return s;
}
static void cfg_to_c(const p25::P3Config& pc, p25_p3_config* o) {
  o->log_blowup = pc.fri_config.log_blowup;
  o->num_queries = pc.fri_config.num_queries;
  o->proof_of_work_bits = pc.fri_config.proof_of_work_bits;
  o->log_quotient_degree = pc.log_quotient_degree;
  o->log_trace_height = pc.log_trace_height;
  o->trace_width = pc.trace_width;
  o->opening_matrix_log_max_height = pc.opening_matrix_log_max_height;
  o->quotient_opened_len = pc.opening_proof_query_openings_opened_values_length;
  o->degree_bits = pc.degree_bits;
}
static p25::P3Config cfg_from_c(const p25_p3_config* c) {
  p25::P3Config pc;
  pc.fri_config.log_blowup = c->log_blowup;
  pc.fri_config.num_queries = c->num_queries;
  pc.fri_config.proof_of_work_bits = c->proof_of_work_bits;
  pc.log_quotient_degree = c->log_quotient_degree;
  pc.log_trace_height = c->log_trace_height;
  pc.trace_width = c->trace_width;
  pc.opening_matrix_log_max_height = c->opening_matrix_log_max_height;
  pc.opening_proof_query_openings_opened_values_length = c->quotient_opened_len;
  pc.degree_bits = c->degree_bits;
  return pc;
}

p25_status p25_p3_prove_fibonacci(int32_t log_n, int32_t num_queries, int32_t pow_bits, uint64_t pow_start,
                                  int32_t threads, uint64_t* inputs_out, size_t cap, size_t* n_out,
                                  p25_p3_config* cfg_out) {
  return host_guarded([&]() -> p25_status {
    if (!n_out) throw std::invalid_argument("null argument");
    p25::P3ProveParams prm;
    prm.log_n = log_n;
    prm.num_queries = num_queries;
    prm.pow_bits = pow_bits;
    prm.pow_start = pow_start;
    prm.threads = threads < 1 ? 1 : threads;
    p25::P3Config pc;
    pc.fri_config.num_queries = num_queries;
    pc.log_trace_height = log_n;
    pc.opening_matrix_log_max_height = log_n + 1;
    pc.degree_bits = log_n;
    if (!inputs_out) {  // size query only
      if (log_n < 1 || log_n > 22 || num_queries < 1) throw std::invalid_argument("bad parameters");
      *n_out = pc.num_inputs();
      if (cfg_out) {
        pc.fri_config.proof_of_work_bits = pow_bits;
        cfg_to_c(pc, cfg_out);
      }
      return P25_OK;
    }
    std::vector<u64> v = p25::p3_prove_fibonacci(prm, pc);
    *n_out = v.size();
    if (cap < v.size()) throw std::invalid_argument("buffer too small");
    memcpy(inputs_out, v.data(), v.size() * 8);
    if (cfg_out) cfg_to_c(pc, cfg_out);
    return P25_OK;
  });
}
p25_status p25_p3_prove_air(const p25_air* air, const uint64_t* trace, int32_t log_n, int32_t num_queries,
                            int32_t pow_bits, uint64_t pow_start, int32_t threads, uint64_t* inputs_out, size_t cap,
                            size_t* n_out, p25_p3_config* cfg_out) {
  return p25_p3_prove_air_ex(air, trace, log_n, 1, num_queries, pow_bits, pow_start, threads, inputs_out, cap, n_out, cfg_out);
}
p25_status p25_p3_prove_air_ex(const p25_air* air, const uint64_t* trace, int32_t log_n, int32_t log_blowup, int32_t num_queries,
                               int32_t pow_bits, uint64_t pow_start, int32_t threads, uint64_t* inputs_out, size_t cap,
                               size_t* n_out, p25_p3_config* cfg_out) {
  return host_guarded([&]() -> p25_status {
    if (!air || !n_out) throw std::invalid_argument("null argument");
    p25::AirProgram prog = air_from_c(air);
    if (log_blowup < 1 || log_blowup > 4) throw std::invalid_argument("log_blowup must be 1..4");
    if (prog.log_quotient_degree() > log_blowup)
      throw std::invalid_argument("an AIR of constraint degree " + std::to_string(prog.max_constraint_degree()) + " needs log_blowup >= " +
                                  std::to_string(prog.log_quotient_degree()));
    p25::P3ProveParams prm;
    prm.log_n = log_n;
    prm.log_blowup = log_blowup;
    prm.num_queries = num_queries;
    prm.pow_bits = pow_bits;
    prm.pow_start = pow_start;
    prm.threads = threads < 1 ? 1 : threads;
    p25::P3Config pc;
    pc.fri_config.log_blowup = log_blowup;
    pc.fri_config.num_queries = num_queries;
    pc.log_trace_height = log_n;
    pc.trace_width = prog.width;
    pc.opening_matrix_log_max_height = log_n + log_blowup;
    pc.degree_bits = log_n;
    pc.log_quotient_degree = prog.log_quotient_degree();   // the number of quotient chunks follows from the AIR's degree (p3_prove_air)
    if (!inputs_out) {  // size query only
      if (log_n < 1 || log_n > 22 || num_queries < 1) throw std::invalid_argument("bad parameters");
      *n_out = pc.num_inputs();
      if (cfg_out) {
        pc.fri_config.proof_of_work_bits = pow_bits;
        cfg_to_c(pc, cfg_out);
      }
      return P25_OK;
    }
    if (!trace) throw std::invalid_argument("null trace");
    if (log_n < 1 || log_n > 22) throw std::invalid_argument("bad parameters");
    const size_t n = (size_t)1 << log_n;
    std::vector<std::vector<u64>> col(prog.width, std::vector<u64>(n));
    for (size_t r = 0; r < n; r++)
      for (int c = 0; c < prog.width; c++) col[c][r] = trace[r * prog.width + c];
    std::vector<u64> v;
    try {
      v = p25::p3_prove_air(prog, col, prm, pc);
    } catch (const std::logic_error& e) {  // "quotient identity does not hold": the trace violates the AIR
      throw std::invalid_argument(e.what());
    }
    *n_out = v.size();
    if (cap < v.size()) throw std::invalid_argument("buffer too small");
    memcpy(inputs_out, v.data(), v.size() * 8);
    if (cfg_out) cfg_to_c(pc, cfg_out);
    return P25_OK;
  });
}
p25_status p25_p3_inputs_to_json(const uint64_t* inputs, size_t n, const p25_p3_config* cfg, char* buf,
                                 size_t cap, size_t* len_out) {
  return host_guarded([&]() -> p25_status {
    if (!inputs || !cfg || !len_out) throw std::invalid_argument("null argument");
    std::string s = p25::p3_inputs_to_json(std::vector<u64>(inputs, inputs + n), cfg_from_c(cfg));
    *len_out = s.size();
    if (buf) {
      if (cap < s.size()) throw std::invalid_argument("buffer too small");
      memcpy(buf, s.data(), s.size());
    }
    return P25_OK;
  });
}

p25_status p25_proof_to_json(p25_circuit* c, const uint64_t* proof, char* buf, size_t cap, size_t* len_out) {
  return host_guarded([&]() -> p25_status {
    if (!c || !proof || !len_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    const p25::Circuit& k = c->c();
    std::string s = p25::proof_to_json(k, p25::make_proof_layout(k), proof);
    *len_out = s.size();
    if (buf) {
      if (cap < s.size()) throw std::invalid_argument("buffer too small");
      memcpy(buf, s.data(), s.size());
    }
    return P25_OK;
  });
}

p25_status p25_proof_to_bytes(p25_circuit* c, const uint64_t* proof, uint8_t* buf, size_t cap, size_t* len_out) {
  return host_guarded([&]() -> p25_status {
    if (!c || !proof || !len_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    const p25::Circuit& k = c->c();
    std::vector<uint8_t> b = p25::proof_to_bytes(k, p25::make_proof_layout(k), proof);
    *len_out = b.size();
    if (buf) {
      if (cap < b.size()) throw std::invalid_argument("buffer too small");
      memcpy(buf, b.data(), b.size());
    }
    return P25_OK;
  });
}
p25_status p25_proof_from_bytes(p25_circuit* c, const uint8_t* bytes, size_t len, uint64_t* proof_out, size_t cap_words) {
  return host_guarded([&]() -> p25_status {
    if (!c || !bytes || !proof_out) throw std::invalid_argument("null argument");
    P25_LOCK(c);
    const p25::Circuit& k = c->c();
    p25::ProofLayout L = p25::make_proof_layout(k);
    if (cap_words < L.total) throw std::invalid_argument("buffer too small");
    p25::proof_from_bytes(k, L, bytes, len, proof_out);
    return P25_OK;
  });
}

}  // extern "C"
