// Poseidon2 permutation over Goldilocks, width 12, x^7, R_F = 8, R_P = 22.
// Follows /root/reference/src/common/poseidon2/poseidon2.rs:59-91 (`Poseidon2::poseidon2`):
//   matmul_external; 4 x {add RC, S-box all, matmul_external}; 22 x {s0 += RC_MID, s0^7,
//   matmul_internal}; 4 x {add RC, S-box all, matmul_external}.
// matmul_external (poseidon2.rs:126-147) = M4 on each 4-block (poseidon2.rs:184-213) then add the
// column sums; matmul_internal (poseidon2.rs:163-182) = x_i * (MAT_DIAG_M_1[i] - 1) + sum(x).
// Constants: poseidon2_goldilocks.rs:10-165 (poseidon2_constants.inc).
// This is the in-circuit hash of the plonky3 verifier (Poseidon2Gate rows); it is used by the
// witness generator (poseidon2_gate.rs:447-523) and the quotient evaluator (poseidon2_gate.rs:233-310).
#pragma once
#include "gl.h"

namespace poseidon2 {

#include "poseidon2_constants.inc"

constexpr int WIDTH = 12;
constexpr int ROUND_F_BEGIN = 4;
constexpr int ROUND_F_END = 8;
constexpr int ROUND_P = 22;
// trace layout (= Poseidon2Gate S-box-input wire order, poseidon2_gate.rs:104-138):
//   [0,36)  full rounds 1..3 (round 0 inputs are not stored), 12 each
//   [36,58) partial rounds
//   [58,106) full rounds 4..7, 12 each
constexpr int TRACE_LEN = 36 + 22 + 48;

GL_HD u64 sbox(u64 x) {  // canonical result; the intermediate powers may stay non-canonical
  u64 x2 = gl::mul_nc(x, x);
  u64 x4 = gl::mul_nc(x2, x2);
  u64 x3 = gl::mul_nc(x, x2);
  return gl::mul(x3, x4);
}

GL_HD void matmul_m4(u64 s[WIDTH]) {
#pragma unroll
  for (int b = 0; b < 3; b++) {
    u64* x = s + 4 * b;
    u64 t0 = gl::add(x[0], x[1]);
    u64 t1 = gl::add(x[2], x[3]);
    u64 t2 = gl::add(t1, gl::add(x[1], x[1]));
    u64 t3 = gl::add(t0, gl::add(x[3], x[3]));
    u64 t1_2 = gl::add(t1, t1), t0_2 = gl::add(t0, t0);
    u64 t4 = gl::add(t3, gl::add(t1_2, t1_2));
    u64 t5 = gl::add(t2, gl::add(t0_2, t0_2));
    x[0] = gl::add(t3, t5);
    x[1] = t5;
    x[2] = gl::add(t2, t4);
    x[3] = t4;
  }
}

GL_HD void matmul_external(u64 s[WIDTH]) {
  matmul_m4(s);
  u64 st[4];
#pragma unroll
  for (int l = 0; l < 4; l++) st[l] = gl::add(gl::add(s[l], s[4 + l]), s[8 + l]);
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = gl::add(s[i], st[i & 3]);
}

GL_HD void matmul_internal(u64 s[WIDTH]) {
  u64 sum = s[0];
#pragma unroll
  for (int i = 1; i < WIDTH; i++) sum = gl::add(sum, s[i]);
#pragma unroll
  for (int i = 0; i < WIDTH; i++) s[i] = gl::canon(gl::mad_nc(s[i], P2_MAT_DIAG_M_1[i] - 1, sum));
}

// Canonical in/out.  `tr(i, v)` receives the S-box inputs the Poseidon2Gate stores as wires
// (trace layout above); NoTrace ignores them.
struct NoTrace {
  GL_HD void operator()(int, u64) const {}
};
template <class Tracer>
GL_HD void permute_impl(u64 s[WIDTH], Tracer& tr) {
  matmul_external(s);
  for (int r = 0; r < ROUND_F_BEGIN; r++) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::add(s[i], P2_RC[12 * r + i]);
    if (r != 0) {
#pragma unroll
      for (int i = 0; i < WIDTH; i++) tr(12 * (r - 1) + i, s[i]);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = sbox(s[i]);
    matmul_external(s);
  }
  for (int r = 0; r < ROUND_P; r++) {
    s[0] = gl::add(s[0], P2_RC_MID[r]);
    tr(36 + r, s[0]);
    s[0] = sbox(s[0]);
    matmul_internal(s);
  }
  for (int r = ROUND_F_BEGIN; r < ROUND_F_END; r++) {
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = gl::add(s[i], P2_RC[12 * r + i]);
#pragma unroll
    for (int i = 0; i < WIDTH; i++) tr(58 + 12 * (r - ROUND_F_BEGIN) + i, s[i]);
#pragma unroll
    for (int i = 0; i < WIDTH; i++) s[i] = sbox(s[i]);
    matmul_external(s);
  }
}

GL_HD void permute(u64 s[WIDTH]) {
  NoTrace nt;
  permute_impl(s, nt);
}

}  // namespace poseidon2
