/* The MULTI-RANK half of libp25's gather (include/p25.h: p25_comm_init, p25_gather_proofs, p25_comm_barrier, p25_comm_max_f64)
 * on a one-GPU box: every rank of a job is a THREAD of this process, and librccl is replaced by the test double
 * tests/c_abi/fake_rccl.cpp (built as librccl.so.1 and put first on LD_LIBRARY_PATH by tests/test_gpu_c_client.py; RCCL itself
 * refuses two ranks on one device).  What executes for the first time outside an 8-GPU node: which rank sends what to whom, the
 * receive offsets of uneven and EMPTY shards, a root other than rank 0, several gathers in flight on one communicator, the max
 * reduction and the barrier over ranks.  Every gathered word is checked against what its rank wrote; the double's counters prove
 * that it -- not a real librccl -- carried the traffic.  The reference has no counterpart (north_star: "RCCL over xGMI only for the
 * final aggregation step"). */
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "p25.h"

#define MAXR 5
#define STRIDE 37 /* words per "proof": odd on purpose (no alignment beyond 8 bytes may be assumed) */

typedef struct {
  int world, dst, rounds;
  size_t counts[MAXR];
} scenario;

static const scenario SCEN[] = {
    {3, 0, 1, {3, 0, 2}},       /* an empty shard in the middle */
    {3, 2, 2, {2, 2, 2}},       /* root = last rank, two gathers in flight on the communicator */
    {4, 1, 1, {1, 4, 0, 2}},    /* uneven, root in the middle, an empty shard */
    {5, 0, 3, {1, 1, 1, 1, 1}}, /* the roots gather of a sharded aggregation: one proof per rank, three in flight */
    {2, 1, 1, {0, 5}},          /* nothing to receive: every proof is the root's own */
};
#define NSCEN ((int)(sizeof SCEN / sizeof SCEN[0]))

typedef struct {
  int rank;
  const scenario* sc;
  const uint8_t* id;
  int failed;
  char msg[256];
} rank_arg;

static uint64_t word_of(int round, int rank, size_t i, size_t w) { return ((uint64_t)(round + 1) << 56) | ((uint64_t)rank << 40) | ((uint64_t)i << 20) | (uint64_t)w; }
static uint32_t status_of(int round, int rank, size_t i) { return (uint32_t)(1000000 * (round + 1) + 1000 * rank + (int)i); }

#define FAIL(...)                                  \
  do {                                             \
    snprintf(a->msg, sizeof a->msg, __VA_ARGS__);  \
    a->failed = 1;                                 \
    return NULL;                                   \
  } while (0)
#define CK(call)                                                                     \
  do {                                                                               \
    p25_status st_ = (call);                                                         \
    if (st_ != P25_OK) FAIL("%s -> %d: %s", #call, (int)st_, p25_last_error());      \
  } while (0)
#define HIPCK(call)                                                                  \
  do {                                                                               \
    hipError_t e_ = (call);                                                          \
    if (e_ != hipSuccess) FAIL("%s -> %s", #call, hipGetErrorString(e_));            \
  } while (0)

static void* rank_main(void* p) {
  rank_arg* a = (rank_arg*)p;
  const scenario* sc = a->sc;
  const int r = a->rank, W = sc->world;
  p25_comm* comm = NULL;
  CK(p25_comm_init(a->id, r, W, &comm));
  if (p25_comm_rank(comm) != r || p25_comm_world(comm) != W) FAIL("communicator shape");
  size_t total = 0, off = 0;
  for (int q = 0; q < W; q++) {
    if (q < r) off += sc->counts[q];
    total += sc->counts[q];
  }
  (void)off;
  const size_t n = sc->counts[r];
  uint64_t* d_p[3] = {0};
  uint32_t* d_s[3] = {0};
  uint64_t* d_all[3] = {0};
  uint32_t* d_alls[3] = {0};
  uint64_t* h = (uint64_t*)malloc((total + 1) * STRIDE * 8);
  uint32_t* hs = (uint32_t*)malloc((total + 1) * 4);
  for (int k = 0; k < sc->rounds; k++) {
    if (n) {
      HIPCK(hipMalloc((void**)&d_p[k], n * STRIDE * 8));
      HIPCK(hipMalloc((void**)&d_s[k], n * 4));
      for (size_t i = 0; i < n; i++) {
        for (size_t w = 0; w < STRIDE; w++) h[i * STRIDE + w] = word_of(k, r, i, w);
        hs[i] = status_of(k, r, i);
      }
      HIPCK(hipMemcpy(d_p[k], h, n * STRIDE * 8, hipMemcpyHostToDevice));
      HIPCK(hipMemcpy(d_s[k], hs, n * 4, hipMemcpyHostToDevice));
    }
    if (r == sc->dst && total) {
      HIPCK(hipMalloc((void**)&d_all[k], total * STRIDE * 8));
      HIPCK(hipMalloc((void**)&d_alls[k], total * 4));
      HIPCK(hipMemset(d_all[k], 0xEE, total * STRIDE * 8));
      HIPCK(hipMemset(d_alls[k], 0xEE, total * 4));
    }
  }
  HIPCK(hipDeviceSynchronize());
  CK(p25_comm_barrier(comm));
  /* `rounds` gathers enqueued back to back on the communicator's stream, then one synchronisation */
  for (int k = 0; k < sc->rounds; k++)
    CK(p25_gather_proofs(comm, NULL, -1, d_p[k], STRIDE, d_s[k], sc->counts, sc->dst, d_all[k], d_alls[k]));
  CK(p25_comm_sync(comm));
  double t = 10.0 + r;      /* the timing protocol: MAX over ranks */
  CK(p25_comm_max_f64(comm, &t));
  if (t != 10.0 + (W - 1)) FAIL("max over %d ranks gave %f", W, t);
  if (r == sc->dst) {
    for (int k = 0; k < sc->rounds; k++) {
      if (!total) break;
      HIPCK(hipMemcpy(h, d_all[k], total * STRIDE * 8, hipMemcpyDeviceToHost));
      HIPCK(hipMemcpy(hs, d_alls[k], total * 4, hipMemcpyDeviceToHost));
      size_t g = 0;
      for (int q = 0; q < W; q++)
        for (size_t i = 0; i < sc->counts[q]; i++, g++) {
          if (hs[g] != status_of(k, q, i)) FAIL("round %d: status %zu is %u, rank %d proof %zu wrote %u", k, g, hs[g], q, i, status_of(k, q, i));
          for (size_t w = 0; w < STRIDE; w++)
            if (h[g * STRIDE + w] != word_of(k, q, i, w))
              FAIL("round %d: word %zu of gathered proof %zu is %016llx, rank %d proof %zu wrote %016llx", k, w, g,
                   (unsigned long long)h[g * STRIDE + w], q, i, (unsigned long long)word_of(k, q, i, w));
        }
    }
  }
  CK(p25_comm_barrier(comm));
  CK(p25_comm_destroy(comm));
  for (int k = 0; k < 3; k++) {
    (void)hipFree(d_p[k]);
    (void)hipFree(d_s[k]);
    (void)hipFree(d_all[k]);
    (void)hipFree(d_alls[k]);
  }
  free(h);
  free(hs);
  return NULL;
}

int main(void) {
  if (p25_device_init(0) != P25_OK) {
    fprintf(stderr, "p25_device_init: %s\n", p25_last_error());
    return 1;
  }
  unsigned long long sends = 0, recvs = 0;
  for (int s = 0; s < NSCEN; s++) {
    const scenario* sc = &SCEN[s];
    uint8_t id[P25_COMM_ID_BYTES];
    if (p25_comm_unique_id(id) != P25_OK) {
      fprintf(stderr, "p25_comm_unique_id: %s\n", p25_last_error());
      return 1;
    }
    if (memcmp(id, "FAKE-RCCL", 9) != 0) {
      fprintf(stderr, "the library loaded a real librccl, not the test double (LD_LIBRARY_PATH)\n");
      return 1;
    }
    pthread_t th[MAXR];
    rank_arg args[MAXR];
    for (int r = 0; r < sc->world; r++) {
      args[r] = (rank_arg){r, sc, id, 0, {0}};
      if (pthread_create(&th[r], NULL, rank_main, &args[r]) != 0) return 1;
    }
    int bad = 0;
    for (int r = 0; r < sc->world; r++) {
      pthread_join(th[r], NULL);
      if (args[r].failed) {
        fprintf(stderr, "scenario %d rank %d: %s\n", s, r, args[r].msg);
        bad = 1;
      }
    }
    if (bad) return 1;
    for (int q = 0; q < sc->world; q++)
      if (q != sc->dst && sc->counts[q]) {
        sends += 2ull * (unsigned)sc->rounds;      /* proofs + statuses per gather */
        recvs += 2ull * (unsigned)sc->rounds;
      }
  }
  /* the double's own counters: it carried exactly the messages the scenarios imply */
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  void (*stats)(unsigned long long*) = lib ? (void (*)(unsigned long long*))dlsym(lib, "fake_rccl_stats") : NULL;
  if (!stats) {
    fprintf(stderr, "fake_rccl_stats not found: %s\n", dlerror());
    return 1;
  }
  unsigned long long st[5];
  stats(st);
  if (st[1] != sends || st[2] != recvs || st[4] == 0) {
    fprintf(stderr, "double's counters: %llu sends (%llu expected), %llu receives (%llu expected), %llu all-reduces\n", st[1], sends, st[2], recvs, st[4]);
    return 1;
  }
  printf("C GATHER MULTI OK: %d scenarios, %llu sends / %llu receives / %llu bytes through the test double, %llu all-reduces\n", NSCEN, st[1], st[2],
         st[3], st[4]);
  return 0;
}
