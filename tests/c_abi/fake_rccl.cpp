// TEST DOUBLE for librccl (tests/test_gpu_c_client.py builds it as librccl.so.1 and puts it first on LD_LIBRARY_PATH):
// the ten RCCL entry points libp25's communicator uses (plonky2.5_amd/csrc/comm.cpp), with every rank of a "job" a THREAD of one
// process on ONE GPU.  A one-GPU box cannot run RCCL with more than one rank (RCCL refuses two ranks on a device), so without
// this the multi-rank half of p25_gather_proofs -- which rank sends what to whom, the receive offsets of uneven and empty
// shards, a root other than rank 0, the max reduction over ranks -- would first execute on the 8-GPU node.  Semantics kept:
// point-to-point messages between a pair of ranks match in posting order; a grouped operation is issued at ncclGroupEnd;
// everything is ordered on the caller's stream (a receive = wait for the sender's stream, device-to-device copy, and the
// sender's stream waits for the copy before it may reuse the buffer).  Not kept: nothing is asynchronous on the host --
// ncclGroupEnd blocks until the peers have arrived -- and only what libp25 calls is implemented (Send / Recv of any type,
// AllReduce of one double with ncclMax).  This is test infrastructure: nothing in the product links or loads it.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct SendRec {
  const void* ptr;
  size_t bytes;
  hipEvent_t ready;      // recorded on the sender's stream when the send was issued
  hipEvent_t done;       // recorded on the receiver's stream behind the copy
  bool done_recorded = false;
};
struct World {
  int n = 0, joined = 0, left = 0;
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<SendRec*>> box;   // (src, dst) -> messages in posting order
  // one-double max reduction: a generation-counted rendezvous
  int red_arrived = 0, red_generation = 0;
  double red_acc = 0, red_result = 0;
};
std::mutex g_mu;
std::map<std::string, World*> g_worlds;
std::atomic<unsigned long long> g_ids{1}, g_sends{0}, g_recvs{0}, g_send_bytes{0}, g_allreduces{0}, g_inits{0};

struct Op {
  bool send;
  void* ptr;
  size_t bytes;
  int peer;
  ncclComm* comm;
  hipStream_t stream;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}
}  // namespace

struct ncclComm {
  World* w;
  int rank;
};

static ncclResult_t run_ops(std::vector<Op>& ops) {
  std::vector<SendRec*> mine;
  // 1. post the sends
  for (Op& o : ops) {
    if (!o.send) continue;
    World* w = o.comm->w;
    SendRec* r = new SendRec{o.ptr, o.bytes, nullptr, nullptr};
    if (hipEventCreateWithFlags(&r->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&r->done, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(r->ready, o.stream) != hipSuccess)
      return ncclUnhandledCudaError;
    {
      std::lock_guard<std::mutex> l(w->mu);
      w->box[{o.comm->rank, o.peer}].push_back(r);
    }
    w->cv.notify_all();
    mine.push_back(r);
    g_sends++;
    g_send_bytes += o.bytes;
  }
  // 2. the receives, in posting order: the matching message is the oldest one from that peer
  for (Op& o : ops) {
    if (o.send) continue;
    World* w = o.comm->w;
    SendRec* r = nullptr;
    {
      std::unique_lock<std::mutex> l(w->mu);
      auto& q = w->box[{o.peer, o.comm->rank}];
      w->cv.wait(l, [&] { return !q.empty(); });
      r = q.front();
      q.pop_front();
    }
    if (r->bytes != o.bytes) return ncclInvalidArgument;     // a send and its receive must agree on the size
    if (hipStreamWaitEvent(o.stream, r->ready, 0) != hipSuccess ||
        hipMemcpyAsync(o.ptr, r->ptr, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess ||
        hipEventRecord(r->done, o.stream) != hipSuccess)
      return ncclUnhandledCudaError;
    {
      std::lock_guard<std::mutex> l(w->mu);
      r->done_recorded = true;
    }
    w->cv.notify_all();
    g_recvs++;
  }
  // 3. my sends complete (on my stream) when the receiver's copy has
  for (size_t i = 0, k = 0; i < ops.size(); i++) {
    if (!ops[i].send) continue;
    SendRec* r = mine[k++];
    World* w = ops[i].comm->w;
    {
      std::unique_lock<std::mutex> l(w->mu);
      w->cv.wait(l, [&] { return r->done_recorded; });
    }
    if (hipStreamWaitEvent(ops[i].stream, r->done, 0) != hipSuccess) return ncclUnhandledCudaError;
    // the events stay alive until both streams have passed them: released at process exit (a test double)
  }
  return ncclSuccess;
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, sizeof id->internal);
  const unsigned long long v = g_ids++;
  memcpy(id->internal, "FAKE-RCCL", 9);
  memcpy(id->internal + 16, &v, sizeof v);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  World* w;
  {
    std::lock_guard<std::mutex> l(g_mu);
    World*& slot = g_worlds[std::string(id.internal, sizeof id.internal)];
    if (!slot) {
      slot = new World();
      slot->n = nranks;
    }
    w = slot;
  }
  if (w->n != nranks) return ncclInvalidArgument;
  {
    std::unique_lock<std::mutex> l(w->mu);       // like the real call: returns when every rank of the job has joined
    w->joined++;
    w->cv.notify_all();
    w->cv.wait(l, [&] { return w->joined >= w->n; });
  }
  *out = new ncclComm{w, rank};
  g_inits++;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) { return ncclCommDestroy(c); }

ncclResult_t ncclGroupStart() {
  t_depth++;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(t_ops);
  return run_ops(ops);
}

static ncclResult_t p2p(bool send, void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t st) {
  if (!c || peer < 0 || peer >= c->w->n || peer == c->rank || !type_bytes(type) || (count && !buf)) return ncclInvalidArgument;
  Op o{send, buf, count * type_bytes(type), peer, c, st};
  if (t_depth) {
    t_ops.push_back(o);
    return ncclSuccess;
  }
  std::vector<Op> one{o};
  return run_ops(one);
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t st) {
  return p2p(true, const_cast<void*>(buf), count, type, peer, c, st);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t c, hipStream_t st) {
  return p2p(false, buf, count, type, peer, c, st);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
  if (!c || !send || !recv || count != 1 || type != ncclFloat64 || op != ncclMax) return ncclInvalidArgument;
  double v = 0;
  if (hipStreamSynchronize(st) != hipSuccess || hipMemcpy(&v, send, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  World* w = c->w;
  double result;
  {
    std::unique_lock<std::mutex> l(w->mu);
    const int gen = w->red_generation;
    w->red_acc = w->red_arrived == 0 ? v : (v > w->red_acc ? v : w->red_acc);
    if (++w->red_arrived == w->n) {
      w->red_result = w->red_acc;
      w->red_arrived = 0;
      w->red_generation++;
      w->cv.notify_all();
    } else {
      w->cv.wait(l, [&] { return w->red_generation != gen; });
    }
    result = w->red_result;
  }
  if (hipMemcpy(recv, &result, sizeof result, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  g_allreduces++;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t e) {
  switch (e) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument (fake rccl)";
    case ncclInvalidUsage: return "invalid usage (fake rccl)";
    case ncclUnhandledCudaError: return "unhandled HIP error (fake rccl)";
    default: return "error (fake rccl)";
  }
}

// what the test reads to make sure THIS library carried the traffic: {inits, sends, recvs, bytes sent, all-reduces}
void fake_rccl_stats(unsigned long long out[5]) {
  out[0] = g_inits;
  out[1] = g_sends;
  out[2] = g_recvs;
  out[3] = g_send_bytes;
  out[4] = g_allreduces;
}

}  // extern "C"
