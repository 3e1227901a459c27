// Latency forms of the lane-cooperative permutations (coop.h) for a LONE wave.
//
// Measured on MI355X (tools/latbench.hip, profiles/r03_b_latbench.txt): a wave that has its SIMD to itself issues an
// independent VALU instruction every ~6.5 cycles and a dependent one every ~9 (v_mad_u64_u32: 8.3 / 11), `s_nop 1`
// costs 10.5, a ds_bpermute round trip 65 (23 each when four are in flight), a DPP operand costs nothing extra.
// So where ONE permutation per wave is all the parallelism there is (witness levels of a single proof, the top of a
// Merkle tree, the transcript), what counts is the instruction count of the round and its LDS round trips:
//   * the hand-scheduled multiply of gl.h (14 VALU + 4 hazard s_nop, one long carry chain) takes 160 cycles on a
//     lone wave, the compiler's reduce128(a*b, mulhi) 123 -- the S-box uses the latter here;
//   * cross-lane traffic goes through DPP row operations (a 16-lane group IS a DPP row) instead of ds_bpermute;
//   * Poseidon2's partial rounds keep a replica of state word 0 in every lane, so the S-box result needs no
//     broadcast, and the sum over the other eleven words is formed while the S-box chain runs.
// Same field arithmetic, canonical values at the same points: bit-identical results (tools/coopbench.hip checks
// every form against the per-lane permutation on the device; the witness / Merkle / transcript parity tests cover them).
#pragma once
#include "coop.h"

namespace coop {

constexpr int DPP_QUAD(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
constexpr int DPP_ROW_SHL(int n) { return 0x100 + n; }  // lane i <- lane i + n of its 16-lane row
constexpr int DPP_ROW_SHR(int n) { return 0x110 + n; }  // lane i <- lane i - n
constexpr int DPP_ROW_ROR(int n) { return 0x120 + n; }  // lane i <- lane (i - n) mod 16

// dst = lane's source under CTRL where that source exists in the row and the destination bank is enabled, else `old`
// (the host compilation pass only parses these functions).
template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ u32 dpp_upd(u32 old, u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, BANK_MASK, false);
#else
  return old ^ v;
#endif
}
template <int CTRL, int BANK_MASK = 0xF>
__device__ __forceinline__ u32 dpp32(u32 v) {
  return dpp_upd<CTRL, BANK_MASK>(v, v);
}
__device__ __forceinline__ u64 mk64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 v) {
  return mk64(dpp32<CTRL>((u32)v), dpp32<CTRL>((u32)(v >> 32)));
}

__device__ __forceinline__ u64 mul_lat(u64 a, u64 b) { return gl::reduce128(a * b, gl::mulhi64(a, b)); }
__device__ __forceinline__ u64 sbox_lat(u64 x) {  // x^7, non-canonical result; x3 and x4 are independent
  u64 x2 = mul_lat(x, x);
  u64 x4 = mul_lat(x2, x2);
  u64 x3 = mul_lat(x, x2);
  return mul_lat(x3, x4);
}

// Sum over the 16 lanes of the row, in every lane (canonical in / out).
__device__ __forceinline__ u64 row_sum(u64 v) {
  v = gl::add(v, dpp64<DPP_ROW_ROR(8)>(v));
  v = gl::add(v, dpp64<DPP_ROW_ROR(4)>(v));
  v = gl::add(v, dpp64<DPP_ROW_ROR(2)>(v));
  v = gl::add(v, dpp64<DPP_ROW_ROR(1)>(v));
  return v;
}
// Lane 0 of the row, in every lane.
__device__ __forceinline__ u64 row_bcast0(u64 v) {
  u32 lo = dpp32<DPP_QUAD(0, 0, 0, 0)>((u32)v), hi = dpp32<DPP_QUAD(0, 0, 0, 0)>((u32)(v >> 32));  // lanes 0..3
  // lanes 4..7 <- 0..3, then 8..15 <- 0..7 (bank masks select the destination quads; invalid sources keep `old`)
  lo = dpp_upd<DPP_ROW_SHR(4), 0x2>(lo, lo);
  hi = dpp_upd<DPP_ROW_SHR(4), 0x2>(hi, hi);
  lo = dpp_upd<DPP_ROW_SHR(8), 0xC>(lo, lo);
  hi = dpp_upd<DPP_ROW_SHR(8), 0xC>(hi, hi);
  return mk64(lo, hi);
}

// Poseidon2 external layer (poseidon2.rs:126-147, 184-213) on a row whose lanes 12..15 hold ZERO:
// out[4b+i] = sum_k M4[i][k] (s[4b+k] + t[k]), t[k] = s[k] + s[4+k] + s[8+k]; M4 = (5 7 1 3 / 4 6 1 1 / 1 3 5 7 / 1 1 4 6).
// The column sums are two rotate-and-add steps over the four quads (the fourth is the zero block), M4 is eight
// multiply-adds by per-lane coefficients on quad-broadcast halves.  Canonical in / out; lanes 12..15 return junk.
__device__ __forceinline__ u64 p2_external_lat(u64 s, int rr) {
  u64 a = gl::add(s, dpp64<DPP_ROW_ROR(8)>(s));
  u64 t = gl::add(a, dpp64<DPP_ROW_ROR(4)>(a));
  u64 u = gl::add(s, t);
  const u32 sh = 8u * (u32)(rr & 3);
  const u32 c0 = (0x01010405u >> sh) & 0xFFu, c1 = (0x01030607u >> sh) & 0xFFu, c2 = (0x04050101u >> sh) & 0xFFu,
            c3 = (0x06070103u >> sh) & 0xFFu;
  const u32 lo = (u32)u, hi = (u32)(u >> 32);
  u64 al = (u64)dpp32<DPP_QUAD(0, 0, 0, 0)>(lo) * c0, ah = (u64)dpp32<DPP_QUAD(0, 0, 0, 0)>(hi) * c0;
  al += (u64)dpp32<DPP_QUAD(1, 1, 1, 1)>(lo) * c1;
  ah += (u64)dpp32<DPP_QUAD(1, 1, 1, 1)>(hi) * c1;
  al += (u64)dpp32<DPP_QUAD(2, 2, 2, 2)>(lo) * c2;
  ah += (u64)dpp32<DPP_QUAD(2, 2, 2, 2)>(hi) * c2;
  al += (u64)dpp32<DPP_QUAD(3, 3, 3, 3)>(lo) * c3;
  ah += (u64)dpp32<DPP_QUAD(3, 3, 3, 3)>(hi) * c3;
  return gl::canon(reduce_row(al, ah));
}

// Poseidon2 with the gate's S-box-input trace (same contract as coop::poseidon2_permute).
template <class Emit>
__device__ inline u64 poseidon2_permute_lat(u64 s, int lane, const u64* __restrict__ k /*LDS, stage_poseidon2_rc*/, Emit emit) {
  using namespace poseidon2;
  const int rr = lane & (GROUP - 1);
  const bool active = rr < 12;
  const int r = active ? rr : 0;
  if (!active) s = 0;
  const u64 diag_m1 = k[118 + r], d0 = k[118];
  s = p2_external_lat(s, rr);
  for (int rd = 0; rd < ROUND_F_BEGIN; rd++) {
    s = gl::add(s, k[12 * rd + r]);
    if (rd != 0 && active) emit(12 * (rd - 1) + r, s);
    s = active ? gl::canon(sbox_lat(s)) : 0;
    s = p2_external_lat(s, rr);
  }
  // partial rounds: z = state word 0, replicated; s = this lane's word (lane 0's copy is dead until the end)
  u64 z = row_bcast0(s);
  if (!active || rr == 0) s = 0;
  for (int rd = 0; rd < ROUND_P; rd++) {
    const u64 zz = gl::add(z, k[96 + rd]);
    if (rr == 0) emit(36 + rd, zz);
    const u64 others = row_sum(s);                       // words 1..11: independent of the S-box chain
    const u64 mine = gl::canon(mul_lat(s, diag_m1));
    const u64 sb = gl::canon(sbox_lat(zz));
    const u64 sum = gl::add(others, sb);
    z = gl::add(gl::canon(mul_lat(sb, d0)), sum);
    s = (active && rr != 0) ? gl::add(mine, sum) : 0;
  }
  if (rr == 0) s = z;
  for (int rd = ROUND_F_BEGIN; rd < ROUND_F_END; rd++) {
    s = gl::add(s, k[12 * rd + r]);
    if (active) emit(58 + 12 * (rd - ROUND_F_BEGIN) + r, s);
    s = active ? gl::canon(sbox_lat(s)) : 0;
    s = p2_external_lat(s, rr);
  }
  return s;
}

// Poseidon (v1) MDS gather through DPP.  State word r in lane r of the row, lanes 12..15 MIRROR words 0..3, so
// "word (r + i) mod 12" is lane r + i whenever r + i <= 15 (one row_shl) and lane r + i - 12 otherwise (one
// row_shr by 12 - i, which overwrites exactly the lanes whose first source was out of the row).
template <int I>
__device__ __forceinline__ u32 mds_src(u32 v) {
  if constexpr (I == 0) {
    return v;
  } else if constexpr (I <= 4) {
    return dpp32<DPP_ROW_SHL(I)>(v);            // r + I <= 15 for every r <= 11
  } else {
    u32 a = dpp32<DPP_ROW_SHL(I)>(v);           // valid for r <= 15 - I; the others keep v (overwritten next)
    return dpp_upd<DPP_ROW_SHR(12 - I)>(a, v);  // r >= 12 - I
  }
}
template <int I>
__device__ __forceinline__ void mds_term(u64& al, u64& ah, u32 lo, u32 hi) {
  al += (u64)mds_src<I>(lo) * poseidon::MDS_CIRC[I];
  ah += (u64)mds_src<I>(hi) * poseidon::MDS_CIRC[I];
}
__device__ __forceinline__ u64 mirror(u64 s) {  // lanes 12..15 <- lanes 0..3
  const u32 lo = (u32)s, hi = (u32)(s >> 32);
  return mk64(dpp_upd<DPP_ROW_SHR(12), 0x8>(lo, lo), dpp_upd<DPP_ROW_SHR(12), 0x8>(hi, hi));
}
template <bool TRACE, class Emit>
__device__ inline u64 poseidon_permute_lat_impl(u64 s, int lane, const u64* __restrict__ rc, Emit emit) {
  const int rr = lane & (GROUP - 1);
  const bool active = rr < 12;
  const int r = active ? rr : rr - 12;   // lanes 12..15 shadow words 0..3 (their results are never used)
  s = poseidon::add_rc(s, rc[r]);
  for (int rd = 0; rd < poseidon::N_ROUNDS; rd++) {
    const bool full = rd < poseidon::HALF_FULL || rd >= poseidon::HALF_FULL + poseidon::N_PARTIAL;
    if constexpr (TRACE) {
      s = gl::canon(s);
      if (full) {
        if (rd != 0 && active) emit((rd < poseidon::HALF_FULL ? 12 * (rd - 1) : 58 + 12 * (rd - poseidon::HALF_FULL - poseidon::N_PARTIAL)) + r, s);
      } else if (rr == 0) {
        emit(36 + (rd - poseidon::HALF_FULL), s);
      }
    }
    const u64 sb = sbox_lat(s);
    s = (full || r == 0) ? sb : s;
    s = mirror(s);
    const u64 c = rd + 1 < poseidon::N_ROUNDS ? rc[12 * (rd + 1) + r] : 0;
    const u32 lo = (u32)s, hi = (u32)(s >> 32);
    u64 al = (u32)c, ah = c >> 32;
    mds_term<0>(al, ah, lo, hi);  mds_term<1>(al, ah, lo, hi);  mds_term<2>(al, ah, lo, hi);
    mds_term<3>(al, ah, lo, hi);  mds_term<4>(al, ah, lo, hi);  mds_term<5>(al, ah, lo, hi);
    mds_term<6>(al, ah, lo, hi);  mds_term<7>(al, ah, lo, hi);  mds_term<8>(al, ah, lo, hi);
    mds_term<9>(al, ah, lo, hi);  mds_term<10>(al, ah, lo, hi); mds_term<11>(al, ah, lo, hi);
    if (r == 0) {
      al += (u64)lo * poseidon::MDS_DIAG0;
      ah += (u64)hi * poseidon::MDS_DIAG0;
    }
    s = reduce_row(al, ah);
  }
  return gl::canon(s);
}
__device__ inline u64 poseidon_permute_lat(u64 s, int lane, const u64* __restrict__ rc) {
  return poseidon_permute_lat_impl<false>(s, lane, rc, [](int, u64) {});
}
template <class Emit>
__device__ inline u64 poseidon_permute_trace_lat(u64 s, int lane, const u64* __restrict__ rc, Emit emit) {
  return poseidon_permute_lat_impl<true>(s, lane, rc, emit);
}

// Single-state form (the transcript): SGPR broadcasts as in coop::poseidon_permute_single, the S-box in its latency form.
__device__ inline u64 poseidon_permute_single_lat(u64 s, int lane, const u64* __restrict__ rc) {
  const int r = lane < 12 ? lane : 0;
  u32 coef[12];
#pragma unroll
  for (int j = 0; j < 12; j++)
    coef[j] = poseidon::MDS_CIRC[(j - r + 12) % 12] + ((r == 0 && j == 0) ? poseidon::MDS_DIAG0 : 0);
  s = poseidon::add_rc(s, rc[r]);
  for (int rd = 0; rd < poseidon::N_ROUNDS; rd++) {
    const bool full = rd < poseidon::HALF_FULL || rd >= poseidon::HALF_FULL + poseidon::N_PARTIAL;
    const u64 sb = sbox_lat(s);
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
}  // namespace coop
