/* Plain-C client of libp25 (include/p25.h): the call sequence a host binding performs for
 * `builder.p3_verify_proof(..); let data = builder.build(); data.prove(pw)` (src/p3/mod.rs:239-260).
 * Built and run by tests/test_gpu_c_client.py. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "p25.h"

#define CHECK(call)                                                                    \
  do {                                                                                 \
    p25_status st_ = (call);                                                           \
    if (st_ != P25_OK) {                                                               \
      fprintf(stderr, "%s -> %d: %s\n", #call, (int)st_, p25_last_error());            \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

int main(void) {
  CHECK(p25_device_init(0));
  /* the hot path's input: a native plonky3 proof of fibonacci(2^4) */
  size_t n_in = 0;
  p25_p3_config cfg;
  CHECK(p25_p3_prove_fibonacci(4, 6, 8, 0, 2, NULL, 0, &n_in, &cfg));
  uint64_t* inputs = (uint64_t*)malloc(n_in * sizeof(uint64_t));
  CHECK(p25_p3_prove_fibonacci(4, 6, 8, 0, 2, inputs, n_in, &n_in, &cfg));
  /* the circuit, once per shape */
  p25_circuit* c = NULL;
  CHECK(p25_circuit_build_p3_verifier(&cfg, P25_AIR_FIBONACCI, &c));
  p25_circuit_info_t info;
  CHECK(p25_circuit_info(c, &info));
  /* two proofs of the same statement, then a tampered input */
  const size_t words = (size_t)info.proof_words;
  uint64_t* batch = (uint64_t*)malloc(3 * n_in * sizeof(uint64_t));
  for (int i = 0; i < 3; i++) memcpy(batch + (size_t)i * n_in, inputs, n_in * sizeof(uint64_t));
  batch[2 * n_in + 9] ^= 1;
  uint64_t seeds[3] = {11, 11, 12};
  uint64_t* proofs = (uint64_t*)calloc(3 * words, sizeof(uint64_t));
  p25_status status[3];
  CHECK(p25_prove_batch(c, batch, 3, seeds, proofs, words, status, NULL));
  if (status[0] != P25_OK || status[1] != P25_OK || status[2] != P25_ERR_WITNESS_CONFLICT) {
    fprintf(stderr, "unexpected statuses %d %d %d\n", (int)status[0], (int)status[1], (int)status[2]);
    return 1;
  }
  if (memcmp(proofs, proofs + words, words * sizeof(uint64_t)) != 0) {
    fprintf(stderr, "same input and seed gave different proofs\n");
    return 1;
  }
  size_t json_len = 0;
  CHECK(p25_proof_to_json(c, proofs, NULL, 0, &json_len));
  printf("C CLIENT OK: degree_bits %llu, %llu inputs, %zu proof words, proof.json %zu bytes\n",
         (unsigned long long)info.degree_bits, (unsigned long long)info.num_inputs, words, json_len);
  p25_circuit_destroy(c);
  free(inputs); free(batch); free(proofs);
  return 0;
}
