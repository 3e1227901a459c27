// ORACLE -- test infrastructure only.  Eight Poseidon (v1) hashes at a time on AVX-512 (ref_hash_x8.cpp): the tuned leg of
// bench.py's cpu_baseline.  Same digests as ref_hash.h's scalar functions (the checker).
#pragma once
#include "ref_hash.h"

bool ref_x8_available();                                  // the host CPU has AVX-512 F + DQ
void ref_poseidon_x8(u64 states[8][12]);                  // ref_poseidon on eight states
// hash_no_pad of rows i0 .. i0+7 of a row-major matrix with `width` > 4 words per row
void ref_hash_rows_x8(const u64* leaves, size_t width, size_t i0, RHash out[8]);
// parents[k] = ref_two_to_one(children[2k], children[2k+1]), k < 8
void ref_two_to_one_x8(const RHash* children, RHash parents[8]);

// the vanishing polynomials at coset points i0 .. i0+7 (ref_quotient_x8.cpp): out[k][j] for challenge k, point i0 + j
struct RCircuit;
struct RPolyBatch;
void ref_vanishing_points_x8(const RCircuit& c, const RPolyBatch& constants_sigmas, const RPolyBatch& wires,
                             const RPolyBatch& zs_batch, const u64* betas, const u64* gammas, const u64* alphas,
                             const u64* pih, size_t i0, const u64 x[8], const u64 l0[8], u64 (*out)[8]);
