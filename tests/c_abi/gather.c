/* Plain-C client of libp25's multi-GPU entry points (include/p25.h: p25_comm_*, p25_gather_proofs): the one collective of
 * the path -- "RCCL over xGMI only for the final aggregation step" (north_star) -- driven the way a Rust host would, with no
 * Python and no torch in the process.  One GPU box, so the job is a world of ONE rank: the communicator is a real RCCL
 * communicator (ncclCommInitRank), the barrier / max reduction are real ncclAllReduce calls on the library's side stream, and
 * the gather orders itself behind the proving streams on the device (p25_circuit_mark), two pipelined steps deep:
 *
 *   step k:  p25_prove_batch_dev -> buffer k & 1;  p25_circuit_mark(slot k & 1);  p25_gather_proofs(.., slot k & 1, ..)
 *            (before buffer k & 1 is proved into again: p25_circuit_wait_stream(c, p25_comm_stream(comm)))
 *
 * The gathered proofs must equal, byte for byte, what the host-buffer entry point gives for the same inputs and seeds.
 * With N ranks the same program runs once per GPU with the id handed over by the launcher (argv: rank world idfile).
 * Built and run by tests/test_gpu_c_client.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "p25.h"

#define CHECK(call)                                                                    \
  do {                                                                                 \
    p25_status st_ = (call);                                                           \
    if (st_ != P25_OK) {                                                               \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, p25_last_error());            \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)
#define HIP(call)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s -> %s\n", #call, hipGetErrorString(e_));                     \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

enum { B = 6, STEPS = 3 };

int main(void) {
  /* p25_device_init_ex tells a host whether its hardware-queue request can still have had an effect: here nothing has
   * touched HIP before, so it must not warn */
  p25_status st0 = p25_device_init_ex(0, P25_DEFAULT_HW_QUEUES);
  if (st0 != P25_OK) {
    fprintf(stderr, "p25_device_init_ex -> %d: %s\n", (int)st0, p25_last_error());
    return 1;
  }
  p25_runtime_info_t ri;
  CHECK(p25_runtime_info(&ri));
  if (ri.device_index != 0 || ri.hw_queues_requested != P25_DEFAULT_HW_QUEUES || ri.hw_queues_setting_late ||
      ri.hw_queues_env != P25_DEFAULT_HW_QUEUES) {
    fprintf(stderr, "runtime info: device %d requested %d env %d late %d\n", ri.device_index, ri.hw_queues_requested,
            ri.hw_queues_env, ri.hw_queues_setting_late);
    return 1;
  }

  uint8_t id[P25_COMM_ID_BYTES];
  p25_comm* comm = NULL;
  CHECK(p25_comm_unique_id(id));
  CHECK(p25_comm_init(id, 0, 1, &comm));
  if (p25_comm_rank(comm) != 0 || p25_comm_world(comm) != 1 || !p25_comm_stream(comm)) {
    fprintf(stderr, "communicator shape\n");
    return 1;
  }
  CHECK(p25_comm_barrier(comm));
  double t = 1.25;
  CHECK(p25_comm_max_f64(comm, &t));
  if (t != 1.25) {
    fprintf(stderr, "max over one rank changed the value: %f\n", t);
    return 1;
  }

  p25_circuit* c = NULL;
  CHECK(p25_circuit_build_gadget(1, 0, &c)); /* xor(x, y): three inputs x, y, x ^ y */
  p25_circuit_info_t info;
  CHECK(p25_circuit_info(c, &info));
  const size_t pw = (size_t)info.proof_words;
  if (info.num_inputs != 3) {
    fprintf(stderr, "unexpected input count %llu\n", (unsigned long long)info.num_inputs);
    return 1;
  }
  /* STEPS batches of B witnesses; the LAST witness of the last step is false (status 4 must travel with the gather) */
  uint64_t in[STEPS][B][3], seeds[STEPS][B];
  for (int k = 0; k < STEPS; k++)
    for (int i = 0; i < B; i++) {
      const uint64_t x = 0x9E3779B97F4A7C15ull * (uint64_t)(k * B + i + 1) >> 32, y = 0xC2B2AE3D27D4EB4Full * (uint64_t)(k + 3 * i + 7) >> 32;
      in[k][i][0] = x;
      in[k][i][1] = y;
      in[k][i][2] = x ^ y;
      seeds[k][i] = (uint64_t)(100 * k + i);
    }
  in[STEPS - 1][B - 1][2] ^= 1;

  /* the reference for the bytes: the host-buffer entry point */
  uint64_t* want = (uint64_t*)calloc((size_t)STEPS * B * pw, 8);
  p25_status want_st[STEPS][B];
  for (int k = 0; k < STEPS; k++) CHECK(p25_prove_batch(c, &in[k][0][0], B, seeds[k], want + (size_t)k * B * pw, pw, want_st[k], NULL));

  uint64_t *d_in, *d_seeds, *d_proofs[2], *d_all[STEPS];
  uint32_t *d_st[STEPS], *d_all_st[STEPS];
  HIP(hipMalloc((void**)&d_in, sizeof in));
  HIP(hipMalloc((void**)&d_seeds, sizeof seeds));
  HIP(hipMemcpy(d_in, in, sizeof in, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_seeds, seeds, sizeof seeds, hipMemcpyHostToDevice));
  for (int b = 0; b < 2; b++) HIP(hipMalloc((void**)&d_proofs[b], (size_t)B * pw * 8));
  for (int k = 0; k < STEPS; k++) {
    HIP(hipMalloc((void**)&d_all[k], (size_t)B * pw * 8));
    HIP(hipMalloc((void**)&d_st[k], B * 4));
    HIP(hipMalloc((void**)&d_all_st[k], B * 4));
    HIP(hipMemset(d_all[k], 0xEE, (size_t)B * pw * 8));
    HIP(hipMemset(d_all_st[k], 0xEE, B * 4));
  }
  HIP(hipDeviceSynchronize());

  /* the pipelined loop: nothing here waits on the host */
  const size_t counts[1] = {B};
  for (int k = 0; k < STEPS; k++) {
    const int buf = k & 1;
    if (k >= 2) CHECK(p25_circuit_wait_stream(c, p25_comm_stream(comm))); /* gather k-2 still reads this buffer */
    CHECK(p25_prove_batch_dev(c, d_in + (size_t)k * B * 3, B, d_seeds + (size_t)k * B, d_proofs[buf], pw, d_st[k], NULL));
    CHECK(p25_circuit_mark(c, (uint32_t)(P25_MAX_MARKS - 1 - buf)));
    CHECK(p25_gather_proofs(comm, c, P25_MAX_MARKS - 1 - buf, d_proofs[buf], pw, d_st[k], counts, 0, d_all[k], d_all_st[k]));
  }
  CHECK(p25_comm_barrier(comm)); /* the barrier runs on the communicator's stream: behind every gather */
  CHECK(p25_comm_sync(comm));

  uint64_t* got = (uint64_t*)malloc((size_t)B * pw * 8);
  uint32_t got_st[B];
  for (int k = 0; k < STEPS; k++) {
    HIP(hipMemcpy(got, d_all[k], (size_t)B * pw * 8, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(got_st, d_all_st[k], sizeof got_st, hipMemcpyDeviceToHost));
    for (int i = 0; i < B; i++) {
      const int bad = (k == STEPS - 1 && i == B - 1);
      if ((int)got_st[i] != (int)want_st[k][i] || (int)got_st[i] != (bad ? P25_ERR_WITNESS_CONFLICT : P25_OK)) {
        fprintf(stderr, "step %d proof %d: status %u (host path %d)\n", k, i, got_st[i], (int)want_st[k][i]);
        return 1;
      }
      if (!bad && memcmp(got + (size_t)i * pw, want + ((size_t)k * B + i) * pw, pw * 8) != 0) {
        fprintf(stderr, "step %d proof %d: gathered bytes differ from the host-buffer path\n", k, i);
        return 1;
      }
    }
  }

  /* argument errors come back as statuses, never as crashes */
  const size_t zero_counts[1] = {0};
  if (p25_gather_proofs(comm, c, -1, NULL, pw, NULL, zero_counts, 0, NULL, NULL) != P25_OK) { /* an empty shard is legal */
    fprintf(stderr, "empty gather: %s\n", p25_last_error());
    return 1;
  }
  if (p25_gather_proofs(comm, c, -1, d_proofs[0], pw, d_st[0], counts, 1, d_all[0], d_all_st[0]) != P25_ERR_INVALID_ARG ||
      p25_gather_proofs(comm, c, -1, NULL, pw, d_st[0], counts, 0, d_all[0], d_all_st[0]) != P25_ERR_INVALID_ARG ||
      p25_gather_proofs(NULL, c, -1, d_proofs[0], pw, d_st[0], counts, 0, d_all[0], d_all_st[0]) != P25_ERR_INVALID_ARG ||
      p25_comm_init(id, 1, 1, &comm) != P25_ERR_INVALID_ARG) {
    fprintf(stderr, "a bad argument was accepted\n");
    return 1;
  }
  CHECK(p25_comm_sync(comm));
  CHECK(p25_comm_destroy(comm));
  p25_circuit_destroy(c);
  printf("C GATHER OK: %d pipelined steps of %d proofs gathered over a library-owned RCCL communicator (world of 1)\n", STEPS, B);
  return 0;
}
