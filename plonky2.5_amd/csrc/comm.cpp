// The one collective of the path behind the C ABI: finished proofs (+ statuses) gathered onto one rank over RCCL/xGMI.
//
// north_star: "independent proofs in a batch shard embarrassingly across the 8 GPUs of one node with RCCL over xGMI only for
// the final aggregation step"; SURVEY.md 2.3 row C1 / 8(e).  The reference has no counterpart: `data.prove(pw)`
// (/root/reference/src/p3/mod.rs:260) borrows the circuit immutably, which is what makes the batch shardable with no exchange
// inside a proof.  A Rust host binds these entry points; nothing here needs Python or torch.
//
// librccl is loaded with dlopen at p25_comm_unique_id / p25_comm_init, not linked: hosts that never go multi-GPU do not need
// it, and inside a torch process the already-loaded copy (same soname, librccl.so.1) is the one that is used.
//
// Ordering is on the device throughout: the gather is enqueued on a stream the communicator owns, which first waits for a
// mark of the producing circuit (p25_circuit_mark) -- so the host never blocks between "step k enqueued" and "gather k
// enqueued", and the gather of step k runs underneath step k + 1's proving.
#include <dlfcn.h>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>
#include <rccl/rccl.h>
#include "../../include/p25.h"
#include "kernels.h"

namespace p25 {
extern thread_local std::string g_last_error;
p25_status ensure_device();

namespace {
struct Rccl {
  void* handle = nullptr;
  std::string err;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok() const { return handle != nullptr; }
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      const char* e = dlerror();
      r.err = std::string("librccl could not be loaded: ") + (e ? e : "unknown error");
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.handle, n);
      if (!p && r.err.empty()) r.err = std::string("librccl lacks ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))sym("ncclCommAbort");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.err.empty()) {
      dlclose(r.handle);
      r.handle = nullptr;
    }
  });
  return r;
}

struct RcclError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
void nccl_check(ncclResult_t e, const char* what) {
  if (e == ncclSuccess) return;
  Rccl& r = rccl();
  throw RcclError(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(e) : "RCCL error") + " (" + std::to_string((int)e) + ")");
}
#define P25_NCCL(expr) nccl_check((expr), #expr)

template <class F>
p25_status comm_guarded(F&& f) {
  try {
    p25_status s = ensure_device();
    if (s != P25_OK) return s;
    Rccl& r = rccl();
    if (!r.ok()) {
      g_last_error = r.err;
      return P25_ERR_RCCL;
    }
    return f(r);
  } catch (const RcclError& e) {
    g_last_error = e.what();
    return P25_ERR_RCCL;
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
}  // namespace
}  // namespace p25

using namespace p25;

// The root's own block moves with a copy KERNEL rather than hipMemcpyAsync: a launch is enqueue-only by construction, whatever
// path the runtime would pick for a device-to-device copy on a stream whose head is a not-yet-satisfied event wait.
namespace {
template <class T>
__global__ __launch_bounds__(256) void k_copy_words(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
template <class T>
void copy_words(const T* src, T* dst, size_t n, hipStream_t st) {
  if (!n || src == dst) return;
  const size_t blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_copy_words<T>, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, src, dst, n);
  P25_HIP(hipGetLastError());
}
}  // namespace

struct p25_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;   // every collective of this communicator runs here, in issue order
  u64* d_scratch = nullptr;       // 2 words: barrier / max-reduction operand
  std::mutex mu;                  // RCCL calls on one communicator are issued by one thread at a time
};

static_assert(P25_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "p25.h's id size is RCCL's");

extern "C" {

p25_status p25_comm_unique_id(uint8_t* id_out) {
  return comm_guarded([&](Rccl& r) -> p25_status {
    if (!id_out) throw std::invalid_argument("id_out is null");
    ncclUniqueId id;
    P25_NCCL(r.GetUniqueId(&id));
    memcpy(id_out, id.internal, P25_COMM_ID_BYTES);
    return P25_OK;
  });
}

p25_status p25_comm_init(const uint8_t* id, int32_t rank, int32_t world, p25_comm** out) {
  return comm_guarded([&](Rccl& r) -> p25_status {
    if (!id || !out) throw std::invalid_argument("null argument");
    if (world < 1 || rank < 0 || rank >= world) throw std::invalid_argument("rank / world out of range");
    std::unique_ptr<p25_comm> c(new p25_comm());
    c->rank = rank;
    c->world = world;
    P25_HIP(hipGetDevice(&c->device));   // ensure_device() has put the thread on the device p25_device_init selected
    ncclUniqueId uid;
    memcpy(uid.internal, id, P25_COMM_ID_BYTES);
    P25_NCCL(r.CommInitRank(&c->comm, world, uid, rank));
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&c->d_scratch, 4 * sizeof(u64));
    if (e != hipSuccess) {
      if (c->stream) (void)hipStreamDestroy(c->stream);
      (void)r.CommAbort(c->comm);
      P25_HIP(e);
    }
    *out = c.release();
    return P25_OK;
  });
}

p25_status p25_comm_destroy(p25_comm* c) {
  if (!c) return P25_OK;
  p25_status s = comm_guarded([&](Rccl& r) -> p25_status {
    std::lock_guard<std::mutex> l(c->mu);
    (void)hipStreamSynchronize(c->stream);
    ncclResult_t e = r.CommDestroy(c->comm);
    (void)hipStreamDestroy(c->stream);
    (void)hipFree(c->d_scratch);
    nccl_check(e, "ncclCommDestroy");
    return P25_OK;
  });
  delete c;
  return s;
}

int32_t p25_comm_rank(const p25_comm* c) { return c ? c->rank : -1; }
int32_t p25_comm_world(const p25_comm* c) { return c ? c->world : 0; }
void* p25_comm_stream(p25_comm* c) { return c ? (void*)c->stream : nullptr; }

p25_status p25_comm_sync(p25_comm* c) {
  return comm_guarded([&](Rccl&) -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    P25_HIP(hipStreamSynchronize(c->stream));
    return P25_OK;
  });
}

// max over ranks of *value (every rank receives it); with value == NULL a plain barrier.  Host-synchronous.
p25_status p25_comm_max_f64(p25_comm* c, double* value) {
  return comm_guarded([&](Rccl& r) -> p25_status {
    if (!c) throw std::invalid_argument("null argument");
    std::lock_guard<std::mutex> l(c->mu);
    double v = value ? *value : 0.0;
    P25_HIP(hipMemcpyAsync(c->d_scratch, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
    P25_NCCL(r.AllReduce(c->d_scratch, c->d_scratch + 1, 1, ncclDouble, ncclMax, c->comm, c->stream));
    P25_HIP(hipMemcpyAsync(&v, c->d_scratch + 1, sizeof v, hipMemcpyDeviceToHost, c->stream));
    P25_HIP(hipStreamSynchronize(c->stream));
    if (value) *value = v;
    return P25_OK;
  });
}
p25_status p25_comm_barrier(p25_comm* c) { return p25_comm_max_f64(c, nullptr); }

p25_status p25_gather_proofs(p25_comm* c, p25_circuit* circuit, int32_t mark_slot, const uint64_t* d_proofs,
                             size_t proof_stride_words, const uint32_t* d_status, const size_t* counts, int32_t dst_rank,
                             uint64_t* d_all_proofs, uint32_t* d_all_status) {
  return comm_guarded([&](Rccl& r) -> p25_status {
    if (!c || !counts) throw std::invalid_argument("null argument");
    if (dst_rank < 0 || dst_rank >= c->world) throw std::invalid_argument("dst_rank out of range");
    if (!proof_stride_words || proof_stride_words > ((size_t)1 << 32)) throw std::invalid_argument("proof_stride_words out of range");
    size_t total = 0;
    for (int q = 0; q < c->world; q++) {
      if (counts[q] > ((size_t)1 << 32)) throw std::invalid_argument("a shard of more than 2^32 proofs");
      total += counts[q];
    }
    const size_t n_local = counts[c->rank];
    if (n_local && (!d_proofs || !d_status)) throw std::invalid_argument("null local buffers");
    const bool root = c->rank == dst_rank;
    if (root && total && (!d_all_proofs || !d_all_status)) throw std::invalid_argument("null destination buffers on dst_rank");
    // device-side ordering behind the producing circuit: its mark (taken when the step had been enqueued), or everything
    // requested so far
    if (circuit) {
      const p25_status s = mark_slot >= 0 ? p25_circuit_stream_wait_mark(circuit, (uint32_t)mark_slot, c->stream)
                                          : p25_circuit_stream_join(circuit, c->stream);
      if (s != P25_OK) return s;
    }
    std::lock_guard<std::mutex> l(c->mu);
    if (root) {   // own block: a copy kernel on the same stream
      size_t off = 0;
      for (int q = 0; q < c->rank; q++) off += counts[q];
      copy_words<u64>(d_proofs, d_all_proofs + off * proof_stride_words, n_local * proof_stride_words, c->stream);
      copy_words<uint32_t>(d_status, d_all_status + off, n_local, c->stream);
    }
    if (c->world == 1) return P25_OK;
    // one group: N - 1 receives on dst_rank, one send on every other rank (point-to-point over xGMI: every sender has its
    // own link to the root, so the gather is bound by the root's links in aggregate, not by a ring)
    P25_NCCL(r.GroupStart());
    ncclResult_t first_err = ncclSuccess;
    auto keep = [&](ncclResult_t e) {
      if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;
    };
    if (root) {
      size_t off = 0;
      for (int q = 0; q < c->world; q++) {
        if (q != c->rank && counts[q]) {
          keep(r.Recv(d_all_proofs + off * proof_stride_words, counts[q] * proof_stride_words, ncclUint64, q, c->comm, c->stream));
          keep(r.Recv(d_all_status + off, counts[q], ncclUint32, q, c->comm, c->stream));
        }
        off += counts[q];
      }
    } else if (n_local) {
      keep(r.Send(d_proofs, n_local * proof_stride_words, ncclUint64, dst_rank, c->comm, c->stream));
      keep(r.Send(d_status, n_local, ncclUint32, dst_rank, c->comm, c->stream));
    }
    const ncclResult_t ge = r.GroupEnd();   // always closed, also after a failed call inside the group
    nccl_check(first_err, "ncclSend / ncclRecv");
    nccl_check(ge, "ncclGroupEnd");
    return P25_OK;
  });
}

}  // extern "C"
