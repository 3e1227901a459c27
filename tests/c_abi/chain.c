/* Plain-C client of libp25 chaining two circuits ON THE DEVICE through the C ABI alone (include/p25.h): two leaf proofs
 * are written to a device buffer and an aggregation circuit proves straight on that buffer -- its inputs ARE its children's
 * flat proofs back to back -- ordered by p25_circuit_mark / p25_circuit_wait_mark, with no host synchronisation in
 * between.  The result must be, byte for byte, what the host-buffer entry points give for the same inputs and seeds.
 * (What a Rust host does instead of `builder.verify_proof` circuits proved one after the other with the proofs passed
 * through PartialWitness, src/p3/mod.rs:260.)  Built and run by tests/test_gpu_c_client.py. */
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

int main(void) {
  CHECK(p25_device_init(0));
  p25_circuit *leaf = NULL, *agg = NULL;
  CHECK(p25_circuit_build_gadget(0, 0, &leaf));                     /* and(x, y): three inputs x, y, x & y */
  CHECK(p25_circuit_build_aggregator(leaf, NULL, NULL, 2, &agg));   /* verifies two leaf proofs, commits to them */
  p25_circuit_info_t li, ai;
  CHECK(p25_circuit_info(leaf, &li));
  CHECK(p25_circuit_info(agg, &ai));
  const size_t lw = (size_t)li.proof_words, aw = (size_t)ai.proof_words;
  if (ai.num_inputs != 2 * lw || li.num_inputs != 3) {
    fprintf(stderr, "unexpected shapes: aggregator inputs %llu, leaf proof words %zu\n", (unsigned long long)ai.num_inputs, lw);
    return 1;
  }
  const uint64_t x0 = 0x0123456789ABCDEFull, y0 = 0x0FEDCBA987654321ull, x1 = 0x1111222233334444ull, y1 = 0x00FF00FF00FF00FFull;
  uint64_t in[6] = {x0, y0, x0 & y0, x1, y1, x1 & y1};
  uint64_t seeds[2] = {7, 8}, seed_agg[1] = {9};

  /* ---- host-buffer path: the reference for the bytes */
  uint64_t* h_leaf = (uint64_t*)calloc(2 * lw, 8);
  uint64_t* h_agg = (uint64_t*)calloc(aw, 8);
  p25_status st2[2], st1[1];
  CHECK(p25_prove_batch(leaf, in, 2, seeds, h_leaf, lw, st2, NULL));
  CHECK(p25_prove_batch(agg, h_leaf, 1, seed_agg, h_agg, aw, st1, NULL));
  if (st2[0] != P25_OK || st2[1] != P25_OK || st1[0] != P25_OK) {
    fprintf(stderr, "host path statuses %d %d %d\n", (int)st2[0], (int)st2[1], (int)st1[0]);
    return 1;
  }

  /* ---- device-resident path: nothing synchronises between the two circuits */
  uint64_t *d_in, *d_seeds, *d_leaf, *d_seed_agg, *d_agg;
  uint32_t *d_st2, *d_st1;
  HIP(hipMalloc((void**)&d_in, sizeof in));
  HIP(hipMalloc((void**)&d_seeds, sizeof seeds));
  HIP(hipMalloc((void**)&d_leaf, 2 * lw * 8));
  HIP(hipMalloc((void**)&d_seed_agg, sizeof seed_agg));
  HIP(hipMalloc((void**)&d_agg, aw * 8));
  HIP(hipMalloc((void**)&d_st2, 8));
  HIP(hipMalloc((void**)&d_st1, 8));
  HIP(hipMemcpy(d_in, in, sizeof in, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_seeds, seeds, sizeof seeds, hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_seed_agg, seed_agg, sizeof seed_agg, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 3; rep++) {   /* three times over the same buffers: the second and third reuse them */
    if (rep) CHECK(p25_circuit_wait_mark(leaf, agg, 1));            /* the aggregator has finished reading d_leaf */
    CHECK(p25_prove_batch_dev(leaf, d_in, 2, d_seeds, d_leaf, lw, d_st2, NULL));
    CHECK(p25_circuit_mark(leaf, 0));
    CHECK(p25_circuit_wait_mark(agg, leaf, 0));
    CHECK(p25_prove_batch_dev(agg, d_leaf, 1, d_seed_agg, d_agg, aw, d_st1, NULL));   /* d_inputs = the leaves' d_proofs */
    CHECK(p25_circuit_mark(agg, 1));
  }
  CHECK(p25_circuit_sync(agg));
  CHECK(p25_circuit_sync(leaf));
  uint64_t* back = (uint64_t*)calloc(aw, 8);
  uint32_t dst[3] = {1, 1, 1};
  HIP(hipMemcpy(back, d_agg, aw * 8, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(dst, d_st2, 8, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(dst + 2, d_st1, 4, hipMemcpyDeviceToHost));
  if (dst[0] || dst[1] || dst[2]) {
    fprintf(stderr, "device path statuses %u %u %u\n", dst[0], dst[1], dst[2]);
    return 1;
  }
  if (memcmp(back, h_agg, aw * 8) != 0) {
    fprintf(stderr, "the device-chained aggregate differs from the host-path aggregate\n");
    return 1;
  }
  /* ---- windows of one buffer (p25_prove_batch_dev_windows): three leaves under a 2-ary aggregator -> groups (0, 1) and,
   * right-aligned and overlapping, (1, 2), in ONE batch straight on the leaves' device buffer; the host path proves the two
   * groups from explicit copies and must give the same bytes */
  {
    const uint64_t x2 = 0x5555AAAA5555AAAAull, y2 = 0x0F0F0F0FF0F0F0F0ull;
    uint64_t in3[9] = {x0, y0, x0 & y0, x1, y1, x1 & y1, x2, y2, x2 & y2};
    uint64_t seeds3[3] = {7, 8, 11}, seeds_w[2] = {9, 10};
    uint64_t* h3 = (uint64_t*)calloc(3 * lw, 8);
    uint64_t* groups = (uint64_t*)calloc(4 * lw, 8);
    uint64_t* h_w = (uint64_t*)calloc(2 * aw, 8);
    p25_status st3[3], stw[2];
    CHECK(p25_prove_batch(leaf, in3, 3, seeds3, h3, lw, st3, NULL));
    memcpy(groups, h3, 2 * lw * 8);                      /* leaves 0, 1 */
    memcpy(groups + 2 * lw, h3 + lw, 2 * lw * 8);        /* leaves 1, 2 */
    CHECK(p25_prove_batch(agg, groups, 2, seeds_w, h_w, aw, stw, NULL));
    if (st3[0] || st3[1] || st3[2] || stw[0] || stw[1]) { fprintf(stderr, "windows: host path statuses\n"); return 1; }
    uint64_t *d_in3, *d_s3, *d_l3, *d_sw, *d_w;
    uint32_t* d_stw;
    HIP(hipMalloc((void**)&d_in3, sizeof in3));
    HIP(hipMalloc((void**)&d_s3, sizeof seeds3));
    HIP(hipMalloc((void**)&d_l3, 3 * lw * 8));
    HIP(hipMalloc((void**)&d_sw, sizeof seeds_w));
    HIP(hipMalloc((void**)&d_w, 2 * aw * 8));
    HIP(hipMalloc((void**)&d_stw, 32));
    HIP(hipMemcpy(d_in3, in3, sizeof in3, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_s3, seeds3, sizeof seeds3, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_sw, seeds_w, sizeof seeds_w, hipMemcpyHostToDevice));
    CHECK(p25_prove_batch_dev(leaf, d_in3, 3, d_s3, d_l3, lw, d_stw + 2, NULL));
    CHECK(p25_circuit_mark(leaf, 2));
    CHECK(p25_circuit_wait_mark(agg, leaf, 2));
    CHECK(p25_prove_batch_dev_windows(agg, d_l3, 2 * lw, 1 * lw, 2, d_sw, d_w, aw, d_stw));
    CHECK(p25_circuit_sync(agg));
    CHECK(p25_circuit_sync(leaf));
    uint64_t* back_w = (uint64_t*)calloc(2 * aw, 8);
    uint32_t dstw[2] = {1, 1};
    HIP(hipMemcpy(back_w, d_w, 2 * aw * 8, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(dstw, d_stw, 8, hipMemcpyDeviceToHost));
    if (dstw[0] || dstw[1] || memcmp(back_w, h_w, 2 * aw * 8) != 0) {
      fprintf(stderr, "windows: the batch over windows differs from the host path (statuses %u %u)\n", dstw[0], dstw[1]);
      return 1;
    }
    /* a last window beyond the ones before it, a zero stride: refused */
    if (p25_prove_batch_dev_windows(agg, d_l3, 2 * lw, 3 * lw, 2, d_sw, d_w, aw, d_stw) != P25_ERR_INVALID_ARG ||
        p25_prove_batch_dev_windows(agg, d_l3, 0, 0, 2, d_sw, d_w, aw, d_stw) != P25_ERR_INVALID_ARG) {
      fprintf(stderr, "windows: bad arguments accepted\n");
      return 1;
    }
    hipFree(d_in3); hipFree(d_s3); hipFree(d_l3); hipFree(d_sw); hipFree(d_w); hipFree(d_stw);
    free(h3); free(groups); free(h_w); free(back_w);
  }
  /* argument checks of the ordering entry points */
  if (p25_circuit_mark(leaf, P25_MAX_MARKS) != P25_ERR_INVALID_ARG || p25_circuit_wait_mark(agg, agg, 0) != P25_ERR_INVALID_ARG ||
      p25_circuit_wait_mark(NULL, leaf, 0) != P25_ERR_INVALID_ARG) {
    fprintf(stderr, "ordering entry points accepted bad arguments\n");
    return 1;
  }
  printf("C CHAIN OK: leaf proof %zu words, aggregate %zu words, %llu public inputs, device-chained == host path\n", lw, aw,
         (unsigned long long)ai.num_public_inputs);
  p25_circuit_destroy(agg);
  p25_circuit_destroy(leaf);
  hipFree(d_in); hipFree(d_seeds); hipFree(d_leaf); hipFree(d_seed_agg); hipFree(d_agg); hipFree(d_st2); hipFree(d_st1);
  free(h_leaf); free(h_agg); free(back);
  return 0;
}
