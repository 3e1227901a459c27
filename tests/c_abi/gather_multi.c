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

/* ---- the same with REAL circuits: every rank proves its shard on its own circuit (device-resident, two pipelined steps), marks,
 * and the gather waits for the mark on the device; the root checks every gathered proof against the host-buffer path. ---- */
#define CSTEPS 2
static void gadget_inputs(int rank, int k, size_t i, uint64_t in[3], uint64_t* seed) {
  const uint64_t x = (0x9E3779B97F4A7C15ull * (uint64_t)(97 * rank + 13 * k + (int)i + 1)) >> 32;
  const uint64_t y = (0xC2B2AE3D27D4EB4Full * (uint64_t)(31 * rank + 7 * k + 3 * (int)i + 5)) >> 32;
  in[0] = x;
  in[1] = y;
  in[2] = x ^ y;
  *seed = (uint64_t)(100000 * rank + 100 * k + (int)i);
}
static void* rank_main_circuit(void* p) {
  rank_arg* a = (rank_arg*)p;
  const scenario* sc = a->sc;
  const int r = a->rank, W = sc->world;
  p25_comm* comm = NULL;
  p25_circuit* c = NULL;
  CK(p25_comm_init(a->id, r, W, &comm));
  CK(p25_circuit_build_gadget(1, 0, &c)); /* xor(x, y): inputs x, y, x ^ y */
  p25_circuit_info_t info;
  CK(p25_circuit_info(c, &info));
  const size_t pw = (size_t)info.proof_words, n = sc->counts[r];
  size_t total = 0;
  for (int q = 0; q < W; q++) total += sc->counts[q];
  uint64_t *d_in = NULL, *d_seeds = NULL, *d_p[2] = {0}, *d_all[CSTEPS] = {0};
  uint32_t *d_st[CSTEPS] = {0}, *d_alls[CSTEPS] = {0};
  uint64_t h_in[CSTEPS][8][3], h_seeds[CSTEPS][8];
  if (n > 8) FAIL("shard too large for the test");
  memset(h_in, 0, sizeof h_in);
  memset(h_seeds, 0, sizeof h_seeds);
  for (int k = 0; k < CSTEPS; k++)
    for (size_t i = 0; i < n; i++) gadget_inputs(r, k, i, h_in[k][i], &h_seeds[k][i]);
  HIPCK(hipMalloc((void**)&d_in, sizeof h_in));
  HIPCK(hipMalloc((void**)&d_seeds, sizeof h_seeds));
  HIPCK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice));
  HIPCK(hipMemcpy(d_seeds, h_seeds, sizeof h_seeds, hipMemcpyHostToDevice));
  for (int b = 0; b < 2; b++) HIPCK(hipMalloc((void**)&d_p[b], 8 * pw * 8));
  for (int k = 0; k < CSTEPS; k++) {
    HIPCK(hipMalloc((void**)&d_st[k], 8 * 4));
    if (r == sc->dst) {
      HIPCK(hipMalloc((void**)&d_all[k], total * pw * 8));
      HIPCK(hipMalloc((void**)&d_alls[k], total * 4));
      HIPCK(hipMemset(d_alls[k], 0xEE, total * 4));
    }
  }
  HIPCK(hipDeviceSynchronize());
  CK(p25_comm_barrier(comm));
  for (int k = 0; k < CSTEPS; k++) {   /* nothing in this loop waits on the host */
    const int buf = k & 1;
    if (k >= 2) CK(p25_circuit_wait_stream(c, p25_comm_stream(comm)));
    if (n) CK(p25_prove_batch_dev(c, d_in + (size_t)k * 8 * 3, n, d_seeds + (size_t)k * 8, d_p[buf], pw, d_st[k], NULL));
    CK(p25_circuit_mark(c, (uint32_t)(P25_MAX_MARKS - 1 - buf)));
    CK(p25_gather_proofs(comm, c, P25_MAX_MARKS - 1 - buf, d_p[buf], pw, d_st[k], sc->counts, sc->dst, d_all[k], d_alls[k]));
  }
  CK(p25_comm_sync(comm));
  if (r == sc->dst) {
    uint64_t* got = (uint64_t*)malloc(total * pw * 8);
    uint32_t* got_st = (uint32_t*)malloc(total * 4);
    uint64_t* want = (uint64_t*)malloc(pw * 8);
    for (int k = 0; k < CSTEPS; k++) {
      HIPCK(hipMemcpy(got, d_all[k], total * pw * 8, hipMemcpyDeviceToHost));
      HIPCK(hipMemcpy(got_st, d_alls[k], total * 4, hipMemcpyDeviceToHost));
      size_t g = 0;
      for (int q = 0; q < W; q++)
        for (size_t i = 0; i < sc->counts[q]; i++, g++) {
          uint64_t in[3], seed;
          p25_status st1;
          gadget_inputs(q, k, i, in, &seed);
          CK(p25_prove_batch(c, in, 1, &seed, want, pw, &st1, NULL));
          if (st1 != P25_OK || got_st[g] != 0) FAIL("step %d rank %d proof %zu: statuses %d / %u", k, q, i, (int)st1, got_st[g]);
          if (memcmp(got + g * pw, want, pw * 8) != 0) FAIL("step %d: gathered proof %zu (rank %d, proof %zu) differs from the host path", k, g, q, i);
        }
    }
    free(got);
    free(got_st);
    free(want);
  }
  CK(p25_comm_barrier(comm));
  CK(p25_circuit_sync(c));
  CK(p25_comm_destroy(comm));
  p25_circuit_destroy(c);
  return NULL;
}
static const scenario CSCEN[] = {
    {3, 0, CSTEPS, {2, 3, 1}},
    {4, 3, CSTEPS, {1, 0, 2, 2}},
};
#define NCSCEN ((int)(sizeof CSCEN / sizeof CSCEN[0]))

int main(void) {
  if (p25_device_init(0) != P25_OK) {
    fprintf(stderr, "p25_device_init: %s\n", p25_last_error());
    return 1;
  }
  unsigned long long sends = 0, recvs = 0;
  for (int s = 0; s < NSCEN + NCSCEN; s++) {
    const scenario* sc = s < NSCEN ? &SCEN[s] : &CSCEN[s - NSCEN];
    void* (*body)(void*) = s < NSCEN ? rank_main : rank_main_circuit;
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
      if (pthread_create(&th[r], NULL, body, &args[r]) != 0) return 1;
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
  printf("C GATHER MULTI OK: %d scenarios (%d of them with real circuits, marks and two pipelined steps), %llu sends / %llu receives / %llu bytes through the test double, %llu all-reduces\n", NSCEN + NCSCEN,
         NCSCEN, st[1], st[2], st[3], st[4]);
  return 0;
}
