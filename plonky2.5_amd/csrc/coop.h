// Lane-cooperative Poseidon / Poseidon2 permutations for the LATENCY-bound parts of the prover.
//
// Throughput hashing uses one lane per permutation (kernels_hash.hip).  Where only a handful of
// permutations are available and they form a dependent chain -- the Fiat-Shamir transcript, the top
// levels of every Merkle tree, witness generation of a single proof -- a lane needs ~30k dependent
// instructions (~60-100 us) per permutation.  Here one permutation is spread over a 16-lane group
// (state word r in lane r of the group, lanes 12..15 idle): S-boxes run in parallel and the linear
// layers use cross-lane reads, cutting the chain to ~2k instructions (~10-16 us).  A wave holds 4
// independent groups.  Same arithmetic, same results as the per-lane versions (bit-exact).
#pragma once
#include "poseidon.h"
#include "poseidon2.h"

namespace coop {

constexpr int GROUP = 16;

__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  u32 lo = (u32)v, hi = (u32)(v >> 32);
  lo = __shfl(lo, src);
  hi = __shfl(hi, src);
  return ((u64)hi << 32) | lo;
}

// Round constants are staged in LDS by the calling kernel: a latency-bound kernel runs ONE permutation
// per lane, so 30 dependent global-memory constant fetches (~1 us each when cold) would dominate it.
__device__ __forceinline__ void stage_poseidon_rc(u64* lds_rc /*[360]*/) {
  for (int i = threadIdx.x; i < 360; i += blockDim.x) lds_rc[i] = poseidon::RC[i];
  __syncthreads();
}
constexpr int P2_LDS_WORDS = 96 + 22 + 12;
__device__ __forceinline__ void stage_poseidon2_rc(u64* lds /*[P2_LDS_WORDS]*/) {
  for (int i = threadIdx.x; i < P2_LDS_WORDS; i += blockDim.x)
    lds[i] = i < 96 ? poseidon2::P2_RC[i] : (i < 118 ? poseidon2::P2_RC_MID[i - 96] : poseidon2::P2_MAT_DIAG_M_1[i - 118] - 1);
  __syncthreads();
}

// al + ah * 2^32 mod p: the 5-instruction device sequence of poseidon_p3r.h (the host compilation pass only
// parses these functions).
__device__ __forceinline__ u64 reduce_row(u64 al, u64 ah) {
#if defined(__HIP_DEVICE_COMPILE__)
  return poseidon::p3r::reduce_row(al, ah);
#else
  return al + (ah << 32);
#endif
}

// Poseidon (v1).  `s`: this lane's state word (lanes r >= 12 of the group carry garbage); canonical in/out.
// As in the per-lane form the constants of round rd + 1 enter through the MDS accumulators of round rd,
// and the row is reduced by the 5-instruction sequence of poseidon_p3r.h.
__device__ inline u64 poseidon_permute(u64 s, int lane, const u64* __restrict__ rc) {
  const int base = lane & ~(GROUP - 1);
  const int rr = lane & (GROUP - 1);
  const int r = rr < 12 ? rr : 0;
  s = poseidon::add_rc(s, rc[r]);
  for (int rd = 0; rd < poseidon::N_ROUNDS; rd++) {
    const bool full = rd < poseidon::HALF_FULL || rd >= poseidon::HALF_FULL + poseidon::N_PARTIAL;
    const u64 sb = poseidon::sbox(s);
    s = (full || r == 0) ? sb : s;
    const u64 c = rd + 1 < poseidon::N_ROUNDS ? rc[12 * (rd + 1) + r] : 0;
    u32 lo = (u32)s, hi = (u32)(s >> 32);
    u64 al = (u32)c, ah = c >> 32;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      int src = i + r;
      src = base + (src >= 12 ? src - 12 : src);
      al += (u64)__shfl(lo, src) * poseidon::MDS_CIRC[i];
      ah += (u64)__shfl(hi, src) * poseidon::MDS_CIRC[i];
    }
    if (r == 0) {
      al += (u64)lo * poseidon::MDS_DIAG0;
      ah += (u64)hi * poseidon::MDS_DIAG0;
    }
    s = reduce_row(al, ah);
  }
  return gl::canon(s);
}

// The same with the S-box inputs handed to emit(k, canonical value) by the lane that owns them, in PoseidonGate wire
// order (full rounds 1..3, the 22 partial rounds' lane 0, full rounds 26..29): the witness generator of recursive
// verifier circuits, whose Merkle paths are chains of dependent permutations.
template <class Emit>
__device__ inline u64 poseidon_permute_trace(u64 s, int lane, const u64* __restrict__ rc, Emit emit) {
  const int base = lane & ~(GROUP - 1);
  const int rr = lane & (GROUP - 1);
  const bool active = rr < 12;
  const int r = active ? rr : 0;
  s = poseidon::add_rc(s, rc[r]);
  for (int rd = 0; rd < poseidon::N_ROUNDS; rd++) {
    const bool full = rd < poseidon::HALF_FULL || rd >= poseidon::HALF_FULL + poseidon::N_PARTIAL;
    s = gl::canon(s);
    if (full) {
      if (rd != 0 && active) emit((rd < poseidon::HALF_FULL ? 12 * (rd - 1) : 58 + 12 * (rd - poseidon::HALF_FULL - poseidon::N_PARTIAL)) + r, s);
    } else if (rr == 0) {
      emit(36 + (rd - poseidon::HALF_FULL), s);
    }
    const u64 sb = poseidon::sbox(s);
    s = (full || r == 0) ? sb : s;
    const u64 c = rd + 1 < poseidon::N_ROUNDS ? rc[12 * (rd + 1) + r] : 0;
    u32 lo = (u32)s, hi = (u32)(s >> 32);
    u64 al = (u32)c, ah = c >> 32;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      int src = i + r;
      src = base + (src >= 12 ? src - 12 : src);
      al += (u64)__shfl(lo, src) * poseidon::MDS_CIRC[i];
      ah += (u64)__shfl(hi, src) * poseidon::MDS_CIRC[i];
    }
    if (r == 0) {
      al += (u64)lo * poseidon::MDS_DIAG0;
      ah += (u64)hi * poseidon::MDS_DIAG0;
    }
    s = reduce_row(al, ah);
  }
  return gl::canon(s);
}

// Same permutation when the wave carries ONE state (the transcript): state word r lives in lane r of the
// wave and the MDS layer broadcasts each word through SGPRs (v_readlane) instead of 24 LDS-crossbar
// shuffles -- the multiply-adds then take the word as their scalar operand and a per-lane coefficient.
// A single wave issues its instructions serially, so the shorter sequence is what counts:
// 15.0 -> 12.9 us per permutation (tools/coopbench.hip).
__device__ inline u64 poseidon_permute_single(u64 s, int lane, const u64* __restrict__ rc) {
  const int r = lane < 12 ? lane : 0;
  u32 coef[12];
#pragma unroll
  for (int j = 0; j < 12; j++)
    coef[j] = poseidon::MDS_CIRC[(j - r + 12) % 12] + ((r == 0 && j == 0) ? poseidon::MDS_DIAG0 : 0);
  s = poseidon::add_rc(s, rc[r]);
  for (int rd = 0; rd < poseidon::N_ROUNDS; rd++) {
    const bool full = rd < poseidon::HALF_FULL || rd >= poseidon::HALF_FULL + poseidon::N_PARTIAL;
    const u64 sb = poseidon::sbox(s);
    s = (full || r == 0) ? sb : s;
    const u64 c = rd + 1 < poseidon::N_ROUNDS ? rc[12 * (rd + 1) + r] : 0;
    u32 lo = (u32)s, hi = (u32)(s >> 32);
    u64 al = (u32)c, ah = c >> 32;
#pragma unroll
    for (int j = 0; j < 12; j++) {
      al += (u64)(u32)__builtin_amdgcn_readlane(lo, j) * coef[j];
      ah += (u64)(u32)__builtin_amdgcn_readlane(hi, j) * coef[j];
    }
    s = reduce_row(al, ah);
  }
  return gl::canon(s);
}

// Poseidon2 linear layers across the group (poseidon2.rs:126-147, 163-182, 184-213).
__device__ __forceinline__ u64 p2_external(u64 s, int base, int r) {
  const int blk = base + (r & ~3);
  u64 x0 = shfl64(s, blk), x1 = shfl64(s, blk + 1), x2 = shfl64(s, blk + 2), x3 = shfl64(s, blk + 3);
  u64 t0 = gl::add(x0, x1);
  u64 t1 = gl::add(x2, x3);
  u64 t2 = gl::add(t1, gl::add(x1, x1));
  u64 t3 = gl::add(t0, gl::add(x3, x3));
  u64 t1_2 = gl::add(t1, t1), t0_2 = gl::add(t0, t0);
  u64 t4 = gl::add(t3, gl::add(t1_2, t1_2));
  u64 t5 = gl::add(t2, gl::add(t0_2, t0_2));
  u64 o;
  switch (r & 3) {
    case 0: o = gl::add(t3, t5); break;
    case 1: o = t5; break;
    case 2: o = gl::add(t2, t4); break;
    default: o = t4; break;
  }
  // + column sum over the three blocks
  const int c = r & 3;
  u64 sum = gl::add(gl::add(shfl64(o, base + c), shfl64(o, base + 4 + c)), shfl64(o, base + 8 + c));
  return gl::add(o, sum);
}
__device__ __forceinline__ u64 p2_internal(u64 s, int base, int r, bool active, u64 diag_m1) {
  u64 v = active ? s : 0;
  u64 sum = v;
#pragma unroll
  for (int off = 8; off >= 1; off >>= 1) sum = gl::add(sum, shfl64(sum, (base + ((r + off) & (GROUP - 1)))));
  // after the 4 rotations every lane of the group holds the sum over the 16 lanes (12 active + 4 zeros)
  return gl::add(gl::mul(s, diag_m1), sum);
}

// Poseidon2 with the gate's S-box-input trace: emit(i, v) is called by the lane that owns trace word i.
// `rr` = lane index within the group; lanes >= 12 take part in shuffles only.
template <class Emit>
__device__ inline u64 poseidon2_permute(u64 s, int lane, const u64* __restrict__ k /*LDS, stage_poseidon2_rc*/, Emit emit) {
  using namespace poseidon2;
  const int base = lane & ~(GROUP - 1);
  const int rr = lane & (GROUP - 1);
  const bool active = rr < 12;
  const int r = active ? rr : 0;
  if (!active) s = 0;
  const u64 diag_m1 = k[118 + r];
  s = p2_external(s, base, rr < 12 ? rr : rr - 4);  // idle lanes mirror a valid block (results unused)
  for (int rd = 0; rd < ROUND_F_BEGIN; rd++) {
    s = gl::add(s, k[12 * rd + r]);
    if (rd != 0 && active) emit(12 * (rd - 1) + r, s);
    s = sbox(s);
    s = p2_external(s, base, rr < 12 ? rr : rr - 4);
  }
  for (int rd = 0; rd < ROUND_P; rd++) {
    if (rr == 0) {
      s = gl::add(s, k[96 + rd]);
      emit(36 + rd, s);
    }
    u64 sb = sbox(s);
    if (rr == 0) s = sb;
    s = p2_internal(s, base, rr, active, diag_m1);
  }
  for (int rd = ROUND_F_BEGIN; rd < ROUND_F_END; rd++) {
    s = gl::add(s, k[12 * rd + r]);
    if (active) emit(58 + 12 * (rd - ROUND_F_BEGIN) + r, s);
    s = sbox(s);
    s = p2_external(s, base, rr < 12 ? rr : rr - 4);
  }
  return s;
}

}  // namespace coop
