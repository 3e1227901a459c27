// Device-resident Fiat-Shamir transcript and proof-of-work search.
//
// Replaces upstream plonky2 @ 3de92d9 iop/challenger.rs `Challenger<F, PoseidonHash>` and
// fri/prover.rs `fri_proof_of_work` (reached from /root/reference/src/p3/mod.rs:260).
// Semantics: duplex sponge, rate 8; `observe` buffers inputs and
// duplexes at 8; `get_challenge` duplexes if inputs are pending or no outputs remain, then pops
// from the END of the 8-word output buffer.  PoW: the smallest u64 witness w such that observing
// w and squeezing yields a challenge with >= pow_bits leading zero bits.
//
// The transcript is strictly sequential, so it is latency- not throughput-bound: one wavefront runs
// it, with sponge lane r living in wavefront lane r and the Poseidon permutation computed
// cooperatively -- S-boxes in parallel across 12 lanes, the circulant MDS layer as 12 cross-lane
// reads (ds_bpermute) + multiply-adds per lane.  That cuts a permutation's dependent chain from
// ~1.5k modular multiplies (one lane doing all 12 state words) to ~30 x (4 + MDS), i.e. a few us.
#include "kernels.h"
#include "coop.h"
#include "prover_kernels.h"

namespace p25 {

using coop::shfl64;

struct Sponge {
  u64 state, inb, outb;  // per-lane words (state: lanes 0..11, buffers: lanes 0..7)
  uint32_t n_in, n_out;  // wave-uniform
  int lane;
  const u64* rc;  // round constants in LDS
  __device__ void duplex() {
    if (lane < (int)n_in) state = inb;
    n_in = 0;
    state = coop::poseidon_permute_single(state, lane, rc);
    outb = state;
    n_out = 8;
  }
  __device__ void observe(u64 x) {
    n_out = 0;
    if (lane == (int)n_in) inb = x;
    n_in++;
    if (n_in == 8) duplex();
  }
  __device__ u64 challenge() {
    if (n_in > 0 || n_out == 0) duplex();
    u64 v = shfl64(outb, (int)n_out - 1);
    n_out--;
    return v;
  }
};

__global__ __launch_bounds__(64) void k_transcript(Transcript* tr, int init, const u64* __restrict__ obs,
                                                   uint32_t n_obs, u64* __restrict__ chal_out, uint32_t n_chal) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 rc_lds[360];
  coop::stage_poseidon_rc(rc_lds);
  const int lane = threadIdx.x;
  Sponge sp;
  sp.lane = lane;
  sp.rc = rc_lds;
  if (init) {
    sp.state = 0;
    sp.inb = 0;
    sp.outb = 0;
    sp.n_in = 0;
    sp.n_out = 0;
  } else {
    sp.state = lane < 12 ? tr->state[lane] : 0;
    sp.inb = lane < 8 ? tr->in[lane] : 0;
    sp.outb = lane < 8 ? tr->out[lane] : 0;
    sp.n_in = tr->n_in;
    sp.n_out = tr->n_out;
  }
  for (uint32_t i = 0; i < n_obs; i++) sp.observe(obs[i]);
  for (uint32_t i = 0; i < n_chal; i++) {
    u64 c = sp.challenge();
    if (lane == 0) chal_out[i] = c;
  }
  if (lane < 12) tr->state[lane] = sp.state;
  if (lane < 8) {
    tr->in[lane] = sp.inb;
    tr->out[lane] = sp.outb;
  }
  if (lane == 0) {
    tr->n_in = sp.n_in;
    tr->n_out = sp.n_out;
  }
}

void launch_transcript(Transcript* d_tr, int init, const u64* d_obs, uint32_t n_obs, u64* d_chal_out,
                       uint32_t n_chal, hipStream_t st) {
  hipLaunchKernelGGL(k_transcript, dim3(1), dim3(64), 0, st, d_tr, init, d_obs, n_obs, d_chal_out, n_chal);
}

// One candidate per lane: result = min over candidates in [base, base + count) that satisfy the PoW
// (UINT64_MAX if none).  Skips the whole range if a smaller witness was already found.
__global__ __launch_bounds__(256) void k_pow_search(const Transcript* __restrict__ tr, int pow_bits, u64 base,
                                                    u64* __restrict__ result) {
  if (*result < base) return;
  u64 cand = base + (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = tr->state[i];
  const uint32_t pos = tr->n_in;
  for (uint32_t i = 0; i < pos; i++) s[i] = tr->in[i];
  // witness goes to sponge position `pos` (an invariant of the challenger: pos < 8)
#pragma unroll
  for (int i = 0; i < 8; i++)
    if ((uint32_t)i == pos) s[i] = cand;
  poseidon::permute(s);
  if (__clzll((long long)s[7]) >= pow_bits) atomicMin((unsigned long long*)result, (unsigned long long)cand);
}
__global__ void k_pow_init(u64* result) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN); *result = ~0ull; }

// hash_no_pad of a proof's public inputs (overwrite-mode sponge, rate 8; hash/hashing.rs `hash_n_to_m_no_pad`): state
// word r in lane r, one cooperative permutation per chunk of 8.
__global__ __launch_bounds__(64) void k_public_inputs(const u64* __restrict__ vals, size_t B, uint32_t p,
                                                      const uint32_t* __restrict__ pi_slots, uint32_t n,
                                                      u64* __restrict__ values_out, u64* __restrict__ hash_out) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 rc_lds[360];
  coop::stage_poseidon_rc(rc_lds);
  const int lane = threadIdx.x;
  u64 s = 0;
  for (uint32_t off = 0; off < n; off += 8) {
    if (lane < 8 && off + lane < n) {
      s = vals[(size_t)pi_slots[off + lane] * B + p];
      values_out[off + lane] = s;
    }
    s = coop::poseidon_permute_single(s, lane, rc_lds);
  }
  if (lane < 4) hash_out[lane] = s;
}
void launch_public_inputs(const u64* d_vals, size_t B, uint32_t p, const uint32_t* d_pi_slots, uint32_t n, u64* d_values_out,
                          u64* d_hash_out, hipStream_t st) {
  hipLaunchKernelGGL(k_public_inputs, dim3(1), dim3(64), 0, st, d_vals, B, p, d_pi_slots, n, d_values_out, d_hash_out);
}

void launch_pow_search(const Transcript* d_tr, int pow_bits, u64* d_result, hipStream_t st) {
  hipLaunchKernelGGL(k_pow_init, dim3(1), dim3(1), 0, st, d_result);
  // Expected number of candidates is 2^pow_bits; a window is skipped once a smaller witness is known.
  // Windows grow geometrically -- 1, 1, 2, 4, ... x 2^pow_bits candidates, 2^(pow_bits+6) in total -- so
  // the expected work is ~1.7 x 2^pow_bits permutations (a flat 2^(pow_bits+2) window costs 4 x) and all
  // windows fail with probability e^-64 (reported as P25_ERR_INTERNAL by k_finish).
  const int wb = pow_bits < 12 ? 12 : pow_bits;
  u64 base = 0;
  for (int w = 0; w < 7; w++) {
    const u64 window = (u64)1 << (w == 0 ? wb : wb + w - 1);
    hipLaunchKernelGGL(k_pow_search, dim3((unsigned)(window / 256)), dim3(256), 0, st, d_tr, pow_bits, base, d_result);
    base += window;
  }
}

}  // namespace p25
