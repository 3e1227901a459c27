// Witness generation on the GPU: one launch per level of the static schedule
// (witness_program.h), one lane per (generator, proof).  Values live in a slot-major array
// vals[slot * batch_stride + proof], so lanes of consecutive proofs are coalesced.
//
// Generator bodies restate the reference's `SimpleGenerator::run_once` implementations:
//   Poseidon2Generator           /root/reference/src/common/poseidon2/poseidon2_gate.rs:447-523
//   U32ArithmeticGenerator       src/common/u32/gates/arithmetic_u32.rs:389-439
//   U32InterleaveGenerator       src/common/u32/gates/interleave_u32.rs:305-334
//   UninterleaveToU32Generator   src/common/u32/gates/uninterleave_to_u32.rs:353-390
// and upstream plonky2 @ 3de92d9's ConstantGenerator, RandomValueGenerator (made deterministic:
// SplitMix64 of a per-proof seed), ArithmeticBaseGenerator, MulExtensionGenerator,
// QuotientGeneratorExtension, BaseSplitGenerator<2>, WireSplitGenerator, BaseSumGenerator<2>,
// LowHighGenerator, ExponentiationGenerator.
#include "kernels.h"
#include "poseidon.h"
#include "poseidon2.h"
#include "coop.h"
#include "coop_lat.h"
#include "prover_kernels.h"

namespace p25 {

__device__ __forceinline__ u64 random_fill(u64 seed, u64 k) {
  u64 z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return z >= gl::P ? z - gl::P : z;
}

struct Emitter {
  u64* vals;
  const uint32_t* outs;
  size_t B;
  uint32_t p;
  uint32_t* status;
  __device__ __forceinline__ void operator()(int k, u64 v) const {
    uint32_t a = outs[k];
    size_t idx = (size_t)(a & 0x7FFFFFFFu) * B + p;
    if (a & WIT_CHECK_FLAG) {
      if (vals[idx] != v) set_status(status + p, 4);  // P25_ERR_WITNESS_CONFLICT
    } else {
      vals[idx] = v;
    }
  }
};
struct P2Tracer {
  const Emitter& em;
  __device__ __forceinline__ void operator()(int i, u64 v) const { em(4 + i, v); }
};

__device__ __forceinline__ void witgen_level_body(uint32_t block, const WitGen* __restrict__ gens,
                                                  const uint32_t* __restrict__ args, uint32_t g_begin,
                                                  uint32_t g_count, u64* __restrict__ vals, size_t B,
                                                  uint32_t n_proofs, const u64* __restrict__ seeds,
                                                  uint32_t* __restrict__ status, uint32_t skip_kind,
                                                  const u64* __restrict__ filler, uint32_t n_filler) {
  size_t idx = (size_t)block * blockDim.x + threadIdx.x;
  if (idx >= (size_t)g_count * n_proofs) return;
  const uint32_t gi = g_begin + (uint32_t)(idx / n_proofs);
  const uint32_t p = (uint32_t)(idx % n_proofs);
  const WitGen g = gens[gi];
  if (g.kind == skip_kind) return;  // done cooperatively (k_witgen_level_fused)
  const uint32_t* dep = args + g.arg_off;
  Emitter emit{vals, dep + g.n_deps, B, p, status};
  auto d = [&](int i) -> u64 { return vals[(size_t)dep[i] * B + p]; };
  switch (g.kind) {
    case GEN_CONSTANT:
      emit(0, g.c0);
      break;
    case GEN_RANDOM:
      emit(0, filler ? filler[(size_t)p * n_filler + g.c1] : random_fill(seeds[p], g.aux));
      break;
    case GEN_ARITHMETIC:
      emit(0, gl::add(gl::mul(gl::mul(d(0), d(1)), g.c0), gl::mul(d(2), g.c1)));
      break;
    case GEN_MUL_EXT: {
      gl::E2 r = gl::mul(gl::mul(gl::E2{d(0), d(1)}, gl::E2{d(2), d(3)}), g.c0);
      emit(0, r.a);
      emit(1, r.b);
      break;
    }
    case GEN_QUOTIENT_EXT: {
      gl::E2 r = gl::mul(gl::E2{d(0), d(1)}, gl::inv(gl::E2{d(2), d(3)}));
      emit(0, r.a);
      emit(1, r.b);
      break;
    }
    case GEN_BASE_SPLIT: {
      u64 s = d(0);
      for (int i = 0; i < g.n_outs; i++) {
        emit(i, s & 1);
        s >>= 1;
      }
      break;
    }
    case GEN_WIRE_SPLIT: {
      u64 v = d(0);
      for (int i = 0; i < g.n_outs; i++) {
        emit(i, v & (((u64)1 << 63) - 1));
        v >>= 63;
      }
      break;
    }
    case GEN_BASE_SUM: {
      u64 s = 0;
      for (int i = (int)g.n_deps - 1; i >= 0; i--) s = gl::add(gl::add(s, s), d(i) & 1);
      emit(0, s);
      break;
    }
    case GEN_LOW_HIGH: {
      u64 v = d(0);
      emit(0, v & (((u64)1 << g.aux) - 1));
      emit(1, v >> g.aux);
      break;
    }
    case GEN_EXPONENTIATION: {
      u64 base = d(0);
      int nb = (int)g.n_deps - 1;
      u64 cur = 1;
      for (int i = 0; i < nb; i++) {
        u64 prev = i == 0 ? 1 : gl::mul(cur, cur);
        u64 bit = d(1 + (nb - 1 - i));
        cur = bit ? gl::mul(prev, base) : prev;
        emit(i, cur);
      }
      emit(nb, cur);
      break;
    }
    case GEN_POSEIDON2: {
      u64 st[12];
#pragma unroll
      for (int i = 0; i < 12; i++) st[i] = d(i);
      u64 swap = d(12);
#pragma unroll
      for (int i = 0; i < 4; i++) emit(i, gl::mul(swap, gl::sub(st[i + 4], st[i])));
      if (swap == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          u64 t = st[i];
          st[i] = st[i + 4];
          st[i + 4] = t;
        }
      }
      P2Tracer tr{emit};
      poseidon2::permute_impl(st, tr);
#pragma unroll
      for (int i = 0; i < 12; i++) emit(4 + 106 + i, st[i]);
      break;
    }
    case GEN_ARITH_EXT: {  // upstream ArithmeticExtensionGenerator
      gl::E2 r = gl::add(gl::mul(gl::mul(gl::E2{d(0), d(1)}, gl::E2{d(2), d(3)}), g.c0), gl::mul(gl::E2{d(4), d(5)}, g.c1));
      emit(0, r.a);
      emit(1, r.b);
      break;
    }
    case GEN_POSEIDON: {  // upstream PoseidonGenerator: same wire layout as the Poseidon2 gate cloned from it
      u64 st[12];
      for (int i = 0; i < 12; i++) st[i] = d(i);
      u64 swap = d(12);
      for (int i = 0; i < 4; i++) emit(i, gl::mul(swap, gl::sub(st[i + 4], st[i])));
      if (swap == 1) {
        for (int i = 0; i < 4; i++) {
          u64 t = st[i];
          st[i] = st[i + 4];
          st[i + 4] = t;
        }
      }
      poseidon::permute_naive_trace(st, [&](int k, u64 v) { emit(4 + k, v); });
      for (int i = 0; i < 12; i++) emit(4 + 106 + i, st[i]);
      break;
    }
    case GEN_RANDOM_ACCESS: {  // upstream RandomAccessGenerator::run_once
      const u64 idx = d(0);
      if (idx >= (u64)RA_VEC) {  // upstream: debug_assert!(access_index < vec_size) / out-of-bounds panic
        set_status(status + p, 7);
        break;
      }
      emit(0, d(1 + (int)idx));
      for (int i = 0; i < RA_BITS; i++) emit(1 + i, (idx >> i) & 1);
      break;
    }
    case GEN_REDUCING:
    case GEN_REDUCING_EXT: {  // upstream ReducingGenerator::run_once: acc = acc * alpha + coeff, every accumulator set
      const bool ext = g.kind == GEN_REDUCING_EXT;
      const int nco = ext ? REDX_COEFFS : RED_COEFFS;
      const gl::E2 alpha{d(0), d(1)};
      gl::E2 acc{d(2), d(3)};
      for (int i = 0; i < nco; i++) {
        acc = gl::mul(acc, alpha);
        acc.a = gl::add(acc.a, d(ext ? 4 + 2 * i : 4 + i));
        if (ext) acc.b = gl::add(acc.b, d(5 + 2 * i));
        emit(2 * i, acc.a);
        emit(2 * i + 1, acc.b);
      }
      break;
    }
    case GEN_POSEIDON_MDS:  // upstream PoseidonMdsGenerator::run_once
      for (int r = 0; r < 12; r++)
        for (int c = 0; c < 2; c++) {
          u64 acc = r == 0 ? gl::mul(d(c), (u64)poseidon::MDS_DIAG0) : 0;
          for (int i = 0; i < 12; i++) acc = gl::add(acc, gl::mul(d(2 * ((i + r) % 12) + c), (u64)poseidon::MDS_CIRC[i]));
          emit(2 * r + c, acc);
        }
      break;
    case GEN_COSET_INTERP: {  // upstream InterpolationGenerator::run_once
      const u64 shift = d(0);
      const gl::E2 x = gl::mul(gl::E2{d(CI_W_POINT), d(CI_W_POINT + 1)}, gl::inv(shift));
      emit(0, x.a);
      emit(1, x.b);
      const u64 gen = gl::root_of_unity(4), inv16 = gl::inv(16);
      gl::E2 eval{0, 0}, prod{1, 0};
      u64 xi = 1;
      for (int c = 0; c <= CI_INTER; c++) {
        if (c > 0) {
          emit(2 + 4 * (c - 1), eval.a);
          emit(3 + 4 * (c - 1), eval.b);
          emit(4 + 4 * (c - 1), prod.a);
          emit(5 + 4 * (c - 1), prod.b);
        }
        const int end = c == 0 ? CI_DEGREE : (1 + (CI_DEGREE - 1) * (c + 1) < CI_POINTS ? 1 + (CI_DEGREE - 1) * (c + 1) : CI_POINTS);
        for (int i = c == 0 ? 0 : 1 + (CI_DEGREE - 1) * c; i < end; i++) {
          const gl::E2 v = gl::mul(gl::E2{d(1 + 2 * i), d(2 + 2 * i)}, gl::mul(xi, inv16));
          const gl::E2 term{gl::sub(x.a, xi), x.b};
          eval = gl::add(gl::mul(eval, term), gl::mul(v, prod));
          prod = gl::mul(prod, term);
          xi = gl::mul(xi, gen);
        }
      }
      emit(2 + 4 * CI_INTER, eval.a);
      emit(3 + 4 * CI_INTER, eval.b);
      break;
    }
    case GEN_U32_ARITHMETIC: {
      u64 o = gl::add(gl::mul(d(0), d(1)), d(2));
      u64 hi = o >> 32, lo = o & 0xFFFFFFFFull;
      emit(0, lo);
      emit(1, hi);
      u64 diff = 0xFFFFFFFFull - hi;
      emit(2, diff == 0 ? 0 : gl::inv(diff));
      for (int j = 0; j < 32; j++) {
        emit(3 + j, o & 3);
        o >>= 2;
      }
      break;
    }
    case GEN_U32_INTERLEAVE: {
      u64 x = d(0);
      u64 xi = 0;
      for (int i = 0; i < 32; i++) {
        u64 bit = (x >> (31 - i)) & 1;
        emit(i, bit);
        xi += bit << (2 * (31 - i));
      }
      emit(32, xi);
      break;
    }
    case GEN_U32_UNINTERLEAVE: {
      u64 x = d(0);
      u64 ev = 0, od = 0;
      for (int j = 0; j < 32; j++) {
        int shift = 2 * (31 - j);
        u64 e = (x >> (shift + 1)) & 1, o = (x >> shift) & 1;
        emit(2 * j, e);
        emit(2 * j + 1, o);
        ev += e << (31 - j);
        od += o << (31 - j);
      }
      emit(64, ev);
      emit(65, od);
      break;
    }
    default:
      set_status(status + p, 7);
  }
}

__global__ __launch_bounds__(256) void k_witgen_level(const WitGen* __restrict__ gens,
                                                      const uint32_t* __restrict__ args, uint32_t g_begin,
                                                      uint32_t g_count, u64* __restrict__ vals, size_t B,
                                                      uint32_t n_proofs, const u64* __restrict__ seeds,
                                                      uint32_t* __restrict__ status, uint32_t skip_kind,
                                                      const u64* __restrict__ filler, uint32_t n_filler) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  witgen_level_body(blockIdx.x, gens, args, g_begin, g_count, vals, B, n_proofs, seeds, status, skip_kind, filler,
                    n_filler);
}

// Poseidon2 generators of one level, one 16-lane group per (generator, proof): used for small batches,
// where a level's latency is that of a single permutation (poseidon2_gate.rs:447-523, cooperatively).
// The permutation generators of a level are contiguous in `gens` and their argument lists are laid out back to
// back with a fixed stride (13 dependencies + 122 outputs), so a group finds its slot indices at
// arg_base + k * COOP_ARG_STRIDE without loading its WitGen record first -- one dependent global load less in a
// kernel whose whole duration is a chain of them around one permutation.
constexpr uint32_t COOP_N_DEPS = 13, COOP_N_OUTS = 4 + 106 + 12, COOP_ARG_STRIDE = COOP_N_DEPS + COOP_N_OUTS;
template <bool P2>
__device__ __forceinline__ void witgen_perm_coop_body(uint32_t block, const u64* k_lds, u64* tr_lds,
                                                      const uint32_t* __restrict__ args, uint32_t arg_base,
                                                      uint32_t g_count, u64* __restrict__ vals, size_t B,
                                                      uint32_t n_proofs, uint32_t* __restrict__ status) {
  size_t grp = ((size_t)block * blockDim.x + threadIdx.x) / coop::GROUP;
  const size_t n_groups = (size_t)g_count * n_proofs;
  const bool valid = grp < n_groups;
  if (!valid) grp = n_groups - 1;  // every lane takes part in the cross-lane operations
  const int lane = threadIdx.x & 63, rr = threadIdx.x & (coop::GROUP - 1), base = lane & ~(coop::GROUP - 1);
  const uint32_t p = (uint32_t)(grp % n_proofs);
  const uint32_t* dep = args + arg_base + (size_t)(grp / n_proofs) * COOP_ARG_STRIDE;
  const uint32_t* outs = dep + COOP_N_DEPS;
  // every index this lane will need, requested up front: its input word, the swap flag, and the slots of the
  // outputs it writes afterwards (delta / state word, up to 7 trace words)
  const uint32_t in_slot = dep[rr < 12 ? rr : 0], swap_slot = dep[12];
  uint32_t o_slot[7];
#pragma unroll
  for (int t = 0; t < 7; t++) o_slot[t] = rr + 16 * t < 106 ? outs[4 + rr + 16 * t] : 0;
  const uint32_t o_delta = outs[rr < 4 ? rr : 0], o_state = outs[4 + 106 + (rr < 12 ? rr : 0)];
  auto put = [&](uint32_t a, u64 v) {  // Emitter::operator() with the slot word already in a register
    if (!valid) return;
    const size_t idx = (size_t)(a & 0x7FFFFFFFu) * B + p;
    if (a & WIT_CHECK_FLAG) {
      if (vals[idx] != v) set_status(status + p, 4);  // P25_ERR_WITNESS_CONFLICT
    } else {
      vals[idx] = v;
    }
  };
  u64 s = rr < 12 ? vals[(size_t)in_slot * B + p] : 0;
  const u64 swap = vals[(size_t)swap_slot * B + p];
  __syncthreads();  // the constants staged by the caller (their loads were in flight beside the ones above)
  u64 up = coop::shfl64(s, base + ((rr + 4) & (coop::GROUP - 1)));
  u64 dn = coop::shfl64(s, base + ((rr + 12) & (coop::GROUP - 1)));
  if (rr < 4) put(o_delta, gl::mul(swap, gl::sub(up, s)));
  if (swap == 1) s = rr < 4 ? up : (rr < 8 ? dn : s);
  // The 106 trace words go to LDS while the permutation runs and are written out afterwards: emitting from inside
  // the rounds put a global store and its address arithmetic into every step of the dependent chain.
  u64* tr = tr_lds + (threadIdx.x / coop::GROUP) * 106;
  if constexpr (P2)
    s = coop::poseidon2_permute_lat(s, lane, k_lds, [&](int i, u64 v) { tr[i] = v; });
  else
    s = coop::poseidon_permute_trace_lat(s, lane, k_lds, [&](int i, u64 v) { tr[i] = v; });
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 7; t++)
    if (rr + 16 * t < 106) put(o_slot[t], tr[rr + 16 * t]);
  if (rr < 12) put(o_state, s);
}
// One launch per level for small batches (every launch costs ~12 us of a single proof's latency): blocks
// [0, nb_level) run the per-lane generators of the level, the rest its Poseidon2 generators cooperatively.
__global__ __launch_bounds__(256) void k_witgen_level_fused(const WitGen* __restrict__ gens,
                                                            const uint32_t* __restrict__ args, uint32_t g_begin,
                                                            uint32_t g_count, uint32_t perm_arg_base, uint32_t p2_count,
                                                            uint32_t nb_level, u64* __restrict__ vals, size_t B,
                                                            uint32_t n_proofs, const u64* __restrict__ seeds,
                                                            uint32_t* __restrict__ status,
                                                            const u64* __restrict__ filler, uint32_t n_filler,
                                                            uint32_t coop_kind) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 k_lds[360];  // Poseidon round constants (360) or the Poseidon2 set (coop::P2_LDS_WORDS)
  __shared__ u64 tr_lds[(256 / coop::GROUP) * 106];  // S-box-input traces of the block's 16 permutations
  if (blockIdx.x < nb_level) {
    witgen_level_body(blockIdx.x, gens, args, g_begin, g_count, vals, B, n_proofs, seeds, status, coop_kind, filler, n_filler);
  } else if (coop_kind == GEN_POSEIDON2) {
    for (int i = threadIdx.x; i < coop::P2_LDS_WORDS; i += blockDim.x)   // = coop::stage_poseidon2_rc without its barrier
      k_lds[i] = i < 96 ? poseidon2::P2_RC[i] : (i < 118 ? poseidon2::P2_RC_MID[i - 96] : poseidon2::P2_MAT_DIAG_M_1[i - 118] - 1);
    witgen_perm_coop_body<true>(blockIdx.x - nb_level, k_lds, tr_lds, args, perm_arg_base, p2_count, vals, B, n_proofs, status);
  } else {
    for (int i = threadIdx.x; i < 360; i += blockDim.x) k_lds[i] = poseidon::RC[i];
    witgen_perm_coop_body<false>(blockIdx.x - nb_level, k_lds, tr_lds, args, perm_arg_base, p2_count, vals, B, n_proofs, status);
  }
}

// vals[slot 0] = 0; vals[input_slots[i]] = inputs[p][i]
// A non-canonical input word (>= p) is reduced and the proof flagged P25_ERR_INVALID_ARG: the host-buffer
// entry point rejects such inputs up front, the device-resident one can only find out here.  An input
// connected to an earlier input is compared with it (P25_ERR_WITNESS_CONFLICT on mismatch; upstream panics
// "was set twice with different values") instead of racing it for the slot.  This is the first kernel of a
// proof, so when both happen the smaller code is reported (deterministic within the launch).
// Proof p of the pass (the call's proof p0 + p) reads its n_inputs words at inputs + min((p0 + p) * in_stride, in_max_off):
// in_stride = n_inputs and no bound for a plain [n_proofs][n_inputs] array; windows over a buffer with the last one
// right-aligned for p25_prove_batch_dev_windows (an aggregation level proving on its children's proofs).
__global__ void k_witgen_set_inputs(const u64* __restrict__ inputs, size_t in_stride, size_t in_max_off, size_t p0,
                                    const uint32_t* __restrict__ input_slots,
                                    const uint32_t* __restrict__ input_first, uint32_t n_inputs,
                                    u64* __restrict__ vals, size_t B, uint32_t n_proofs,
                                    uint32_t* __restrict__ status) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)(n_inputs + 1) * n_proofs) return;
  uint32_t i = (uint32_t)(idx / n_proofs), p = (uint32_t)(idx % n_proofs);
  auto flag = [&](uint32_t code) {
    if (atomicCAS(status + p, 0u, code) != 0u) atomicMin(status + p, code);
  };
  if (i == n_inputs) {
    vals[p] = 0;
  } else {
    size_t off = (p0 + p) * in_stride;
    if (off > in_max_off) off = in_max_off;
    u64 v = inputs[off + i];
    if (v >= gl::P) {
      v -= gl::P;
      flag(1);  // P25_ERR_INVALID_ARG
    }
    const uint32_t a = input_slots[i];
    if (a & WIT_CHECK_FLAG) {
      u64 w = inputs[off + input_first[i]];
      if (w >= gl::P) w -= gl::P;
      if (w != v) flag(4);  // P25_ERR_WITNESS_CONFLICT
    } else {
      vals[(size_t)a * B + p] = v;
    }
  }
}

// wires[col][row] (column-major, natural row order) for one proof of the batch
__global__ void k_witgen_fill_wires(const u64* __restrict__ vals, size_t B, uint32_t p,
                                    const uint32_t* __restrict__ wire_slot_cm, size_t n_elems,
                                    u64* __restrict__ wires) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_elems) return;
  wires[e] = vals[(size_t)wire_slot_cm[e] * B + p];
}

void launch_witgen(const DeviceWitnessProgram& wp, const u64* d_inputs, const u64* d_seeds, u64* d_vals,
                   size_t B, uint32_t n_proofs, uint32_t* d_status, hipStream_t st, const u64* d_filler,
                   size_t in_stride, size_t in_max_off, size_t p0) {
  size_t tot = (size_t)(wp.n_inputs + 1) * n_proofs;
  if (in_stride == 0) in_stride = wp.n_inputs;
  hipLaunchKernelGGL(k_witgen_set_inputs, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, d_inputs, in_stride,
                     in_max_off, p0, wp.d_input_slots, wp.d_input_first, wp.n_inputs, d_vals, B, n_proofs, d_status);
  for (size_t l = 0; l + 1 < wp.level_start.size(); l++) {
    uint32_t b = wp.level_start[l], cnt = wp.level_start[l + 1] - b;
    if (!cnt) continue;
    size_t th = (size_t)cnt * n_proofs;
    // The level's permutation generators run cooperatively (16 lanes each, trace staged in LDS), the rest per lane.
    // A pass holds at most 64 proofs and a level ~50 permutations per proof, i.e. fewer waves than the chip has
    // SIMDs either way, so what a level costs is the latency of ONE generator: ~25 us cooperatively against
    // ~480 us for the per-lane form (118 dependent slot-index loads + stores per permutation).
    const uint32_t p2c = l < wp.level_p2_count.size() ? wp.level_p2_count[l] : 0;
    const int coop_p2 = p2c > 0 ? 1 : 0;
    if (!coop_p2) {
      hipLaunchKernelGGL(k_witgen_level, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, st, wp.d_gens, wp.d_args,
                         b, cnt, d_vals, B, n_proofs, d_seeds, d_status, 0xFFFFFFFFu, d_filler, wp.num_random_fill);
    } else {
      const unsigned nb1 = p2c < cnt ? (unsigned)((th + 255) / 256) : 0;
      const size_t th2 = (size_t)p2c * n_proofs * coop::GROUP;
      hipLaunchKernelGGL(k_witgen_level_fused, dim3(nb1 + (unsigned)((th2 + 255) / 256)), dim3(256), 0, st, wp.d_gens,
                         wp.d_args, b, cnt, wp.level_perm_arg_base[l], p2c, nb1, d_vals, B, n_proofs, d_seeds, d_status, d_filler,
                         wp.num_random_fill, wp.level_coop_kind[l]);
    }
  }
}
void launch_fill_wires(const DeviceWitnessProgram& wp, const u64* d_vals, size_t B, uint32_t p, u64* d_wires,
                       hipStream_t st) {
  size_t ne = wp.n_wire_elems;
  hipLaunchKernelGGL(k_witgen_fill_wires, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, d_vals, B, p,
                     wp.d_wire_slot_cm, ne, d_wires);
}

}  // namespace p25
