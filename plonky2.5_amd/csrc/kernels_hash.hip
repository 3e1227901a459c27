// Poseidon-v1 hashing kernels for gfx950: batch permutation, Merkle leaf sponge, 2-to-1 tree levels.
//
// Replaces (upstream plonky2 @ 3de92d9, called from /root/reference/src/p3/mod.rs:260):
//   PoseidonHash::hash_or_noop / two_to_one and MerkleTree::new.
//
// Design: one lane per hash.  A Poseidon state is 12 x u64 = 24 VGPRs; the work per hash
// (>= 1 permutation ~ 1.5k 64-bit modular multiply-equivalents) dwarfs its memory traffic, so the
// kernel is integer-VALU bound and the job of the memory layout is only to stay out of the way:
// leaves are read from a COLUMN-major matrix (column c of leaf l at cols[c * col_stride + l]), so
// the 64 lanes of a wave read 64 consecutive u64 of one column per load -- fully coalesced with no
// transpose pass (upstream transposes the LDE to row-major leaves first; the digest is the same).
// Digests are stored as 4 consecutive u64 per node (32 B per lane, contiguous across lanes).
#include "kernels.h"
#include "poseidon.h"
#include "poseidon2.h"
#include "coop.h"
#include "coop_lat.h"

namespace p25 {

__global__ __launch_bounds__(256) void k_poseidon_permute(u64* states, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
  poseidon::permute(s);
#pragma unroll
  for (int k = 0; k < 12; k++) states[i * 12 + k] = s[k];
}

__global__ __launch_bounds__(256) void k_poseidon2_permute(u64* states, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s[12];
#pragma unroll
  for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
  poseidon2::permute(s);
#pragma unroll
  for (int k = 0; k < 12; k++) states[i * 12 + k] = s[k];
}

// digests[l] = hash_or_noop(leaf l), leaf l = (cols[c*col_stride + l])_{c < width}
// 64-lane workgroups: the kernel is a pure per-lane VALU loop (~200k instructions per lane for 135
// columns), and one wave per workgroup lets the dispatcher back-fill SIMDs as soon as a single wave
// retires (tools/hashbench.hip: 5 % over 256-lane workgroups).
// 80 VGPRs (6 waves per SIMD) leave room for other streams' waves next to it.
__device__ __forceinline__ void hash_leaf(const u64* __restrict__ cols, size_t col_stride, int width,
                                          size_t n_leaves, u64* __restrict__ digests) {
  size_t l = (size_t)blockIdx.x * 64 + threadIdx.x;   // launched with 64 lanes per workgroup
  if (l >= n_leaves) return;
  u64 out[4];
  poseidon::hash_or_noop_strided(cols + l, col_stride, width, out);
  u64* d = digests + 4 * l;
#pragma unroll
  for (int i = 0; i < 4; i++) d[i] = out[i];
}
__global__ __launch_bounds__(64, 6) void k_hash_leaves(const u64* __restrict__ cols, size_t col_stride,
                                                       int width, size_t n_leaves, u64* __restrict__ digests) {
  hash_leaf(cols, col_stride, width, n_leaves, digests);
}
// The same kernel under its own symbol for wide matrices (the 135-column wires LDE: the dominant launch
// of a proof), so that profiler summaries list it separately from the 20- and 16-column commits.
__global__ __launch_bounds__(64, 6) void k_hash_leaves_wide(const u64* __restrict__ cols, size_t col_stride,
                                                            int width, size_t n_leaves, u64* __restrict__ digests) {
  hash_leaf(cols, col_stride, width, n_leaves, digests);
}

// parents[m] = two_to_one(children[2m], children[2m+1])
// Compiled for 8 waves per SIMD: one permutation with a zero capacity needs 57 VGPRs, and the issue rate of this instruction mix
// still rises from 6 to 8 resident waves (profiles/r06_instr_rates.txt: 4.35-4.42 -> 4.28-4.32 cycles per instruction); in the
// pipeline 145.7 against 145.3 proofs/s in every round of three (profiles/r06_ab_hash_occupancy_7_8.txt).  The leaf sponges stay
// at 6 (79 VGPRs: 7 waves = 71 VGPRs and 8 = 63 VGPRs + 28 B of scratch measure the same as 6 there).
__global__ __launch_bounds__(64, 8) void k_tree_level(const u64* __restrict__ children,
                                                    u64* __restrict__ parents, size_t n_parents) {
  size_t m = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (m >= n_parents) return;
  u64 l[4], r[4], o[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    l[i] = children[8 * m + i];
    r[i] = children[8 * m + 4 + i];
  }
  poseidon::two_to_one(l, r, o);
#pragma unroll
  for (int i = 0; i < 4; i++) parents[4 * m + i] = o[i];
}

// Same, one 16-lane group per parent (coop.h): used for the small top levels of a tree, where a level
// is a handful of hashes and its latency, not its throughput, is what the proof waits for.
__global__ __launch_bounds__(256) void k_tree_level_coop(const u64* __restrict__ children,
                                                         u64* __restrict__ parents, size_t n_parents) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 rc_lds[360];
  coop::stage_poseidon_rc(rc_lds);
  size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / coop::GROUP;
  const int rr = threadIdx.x & (coop::GROUP - 1);
  const bool valid = g < n_parents;
  if (!valid) g = n_parents - 1;  // keep every lane in the shuffles
  u64 s = rr < 8 ? children[8 * g + rr] : 0;
  s = coop::poseidon_permute_lat(s, threadIdx.x & 63, rc_lds);
  if (valid && rr < 4) parents[4 * g + rr] = s;
}
// The top of a tree in ONE launch: block b reduces the subtree under cap entry b, from a level with up to 32 nodes
// per cap entry down to the entry itself, one cooperative permutation per parent (16 groups of 16 lanes per block:
// four waves, one per SIMD of the CU -- more waves per SIMD would share its issue slots and stretch every
// permutation, which a first attempt with 64 groups per block showed: +0.15 ms per tree), the level just computed
// staying in LDS for the next one; a wave with no parent left at a level skips it.  A level of this size is a single
// permutation deep whatever the kernel (~14 us), so launching the top levels one by one costs a kernel boundary
// plus the constants' staging per level for nothing.
// `nodes`: the level with `per_block` nodes per cap entry (4 words each, tree layout); the levels above it follow
// in memory (merkle_level_offset), which is where the parents are written.
constexpr int TOP_GROUPS = 16;                       // 256 lanes per block
constexpr size_t TOP_MAX_NODES = 2 * TOP_GROUPS;     // nodes per cap entry at the level the kernel starts from
__global__ __launch_bounds__(256) void k_tree_top_coop(u64* __restrict__ nodes, uint32_t per_block, uint32_t n_blocks) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 rc_lds[360];
  __shared__ u64 lvl[2][TOP_GROUPS * 4];
  coop::stage_poseidon_rc(rc_lds);
  const int g = threadIdx.x / coop::GROUP, rr = threadIdx.x & (coop::GROUP - 1), lane = threadIdx.x & 63;
  const uint32_t b = blockIdx.x;
  u64* cur = nodes;                       // level base (all blocks)
  size_t level_nodes = (size_t)per_block * n_blocks;
  uint32_t np = per_block >> 1;           // parents of this block at the level being computed
  int buf = 0;
  bool first = true;
  while (np >= 1) {
    u64* nxt = cur + 4 * level_nodes;     // next level's base
    const uint32_t j = (uint32_t)g < np ? (uint32_t)g : np - 1;   // idle groups shadow the last parent (results unused)
    if ((uint32_t)(g & ~3) < np) {        // wave-uniform: its first group still has a parent at this level
      u64 s = 0;
      if (rr < 8) s = first ? cur[4 * ((size_t)b * 2 * np) + 8 * j + rr] : lvl[buf ^ 1][8 * j + rr];
      s = coop::poseidon_permute_lat(s, lane, rc_lds);
      if ((uint32_t)g < np && rr < 4) {
        nxt[4 * ((size_t)b * np + j) + rr] = s;
        lvl[buf][4 * j + rr] = s;
      }
    }
    __syncthreads();
    cur = nxt;
    level_nodes >>= 1;
    np >>= 1;
    buf ^= 1;
    first = false;
  }
}

// Levels with at most this many parents use the cooperative per-level kernel (above the fused top).  512 when many
// proofs are in flight (the per-lane form costs several times fewer instructions and other streams hide its latency;
// measured 4096 / 512 / 0: 122.3 / 123.0 / 122.8 proofs/s); a lone proof prefers 32768: at that size the per-lane
// form leaves most SIMDs with one wave or none, and the level takes a full permutation latency.
constexpr size_t COOP_PARENTS_BATCH = 512, COOP_PARENTS_SINGLE = 32768;

static void launch_level(const u64* cur, u64* nxt, size_t m, hipStream_t st, bool single_proof) {
  if (m <= (single_proof ? COOP_PARENTS_SINGLE : COOP_PARENTS_BATCH)) {
    size_t th = m * coop::GROUP;
    hipLaunchKernelGGL(k_tree_level_coop, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, st, cur, nxt, m);
  } else {
    hipLaunchKernelGGL(k_tree_level, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, st, cur, nxt, m);
  }
}
// Levels from `cur` (m nodes) up to the cap: one launch per level while a cap entry has more than TOP_MAX_NODES
// nodes under it, then the fused top.
static u64* launch_levels_to_cap(u64* cur, size_t m, unsigned cap_height, hipStream_t st, bool single_proof) {
  const size_t cap = (size_t)1 << cap_height;
  while (m > cap) {
    if (m / cap <= TOP_MAX_NODES && cap <= 65536) {
      hipLaunchKernelGGL(k_tree_top_coop, dim3((unsigned)cap), dim3(TOP_GROUPS * coop::GROUP), 0, st, cur,
                         (uint32_t)(m / cap), (uint32_t)cap);
      while (m > cap) {
        cur += 4 * m;
        m >>= 1;
      }
      return cur;
    }
    // (Measured and not adopted, round 4: five cooperative levels per launch for a lone proof -- k_tree_top_coop with every
    // block reducing a 32-node subtree -- instead of five launches: 2048 blocks keep eight waves per SIMD resident through
    // the later levels and the wires commit of a lone proof went 5.8 -> 5.96 ms.)
    u64* nxt = cur + 4 * m;
    m >>= 1;
    launch_level(cur, nxt, m, st, single_proof);
    cur = nxt;
  }
  return cur;
}

// Shader clock under the hashing load (bench.py's VALU view prices instructions in cycles): every wave runs
// `reps` permutations and one wave per 1024 blocks reports its elapsed shader cycles (s_memtime) and wall-clock
// ticks (the constant-rate counter, hipDeviceAttributeWallClockRate kHz).
__global__ __launch_bounds__(64) void k_clock_probe(u64* sink, int reps, unsigned long long* clk) {
  size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = l * 12 + i;
  unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int r = 0; r < reps; r++) {
    s[0] ^= (u64)r;
    poseidon::permute(s);
  }
  unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (s[0] == 0x123456789ull) sink[l & 63] = s[1];  // keeps the loop alive
  if (threadIdx.x == 0 && (blockIdx.x & 1023) == 0) {
    clk[2 * (blockIdx.x >> 10)] = c1 - c0;
    clk[2 * (blockIdx.x >> 10) + 1] = w1 - w0;
  }
}
double measure_shader_clock_hz(hipStream_t st) {
  const unsigned blocks = 8192, probes = blocks / 1024;
  u64* d = nullptr;
  P25_HIP(hipMalloc(&d, (64 + 2 * probes) * 8));
  unsigned long long* clk = (unsigned long long*)(d + 64);
  int dev = 0, khz = 0;
  P25_HIP(hipGetDevice(&dev));
  P25_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
  unsigned long long h[2 * probes];
  for (int it = 0; it < 2; it++)  // first pass warms the clocks up
    hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(64), 0, st, d, 64, clk);
  hipError_t e = hipMemcpyAsync(h, clk, sizeof(h), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  P25_HIP(e);
  double cyc = 0, ticks = 0;
  for (unsigned i = 0; i < probes; i++) {
    cyc += (double)h[2 * i];
    ticks += (double)h[2 * i + 1];
  }
  if (ticks <= 0 || khz <= 0) return 0.0;
  return cyc / (ticks / ((double)khz * 1e3));
}

void launch_poseidon_permute(u64* d_states, size_t n, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_poseidon_permute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_states, n);
}
void launch_poseidon2_permute(u64* d_states, size_t n, hipStream_t st) {
  if (!n) return;
  hipLaunchKernelGGL(k_poseidon2_permute, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_states, n);
}

size_t merkle_tree_words(size_t n_leaves, unsigned cap_height) {
  // levels 0 (leaf digests) .. log2(n) - cap_height (the cap), 4 words per node
  size_t total = 0;
  for (size_t m = n_leaves; m >= ((size_t)1 << cap_height); m >>= 1) {
    total += 4 * m;
    if (m == 1) break;
  }
  return total;
}
size_t merkle_level_offset(size_t n_leaves, unsigned level) {
  size_t off = 0;
  size_t m = n_leaves;
  for (unsigned k = 0; k < level; k++) {
    off += 4 * m;
    m >>= 1;
  }
  return off;
}

// Builds every level up to the cap in `tree` (layout: merkle_level_offset).  Returns pointer to cap.
u64* launch_merkle_tree(const u64* d_cols, size_t col_stride, int width, size_t n_leaves,
                        unsigned cap_height, u64* d_tree, hipStream_t st, hipEvent_t ev_begin,
                        hipEvent_t ev_end, bool single_proof) {
  if (ev_begin) (void)hipEventRecord(ev_begin, st);
  if (width >= 128)
    hipLaunchKernelGGL(k_hash_leaves_wide, dim3((unsigned)((n_leaves + 63) / 64)), dim3(64), 0, st, d_cols,
                       col_stride, width, n_leaves, d_tree);
  else
    hipLaunchKernelGGL(k_hash_leaves, dim3((unsigned)((n_leaves + 63) / 64)), dim3(64), 0, st, d_cols,
                       col_stride, width, n_leaves, d_tree);
  if (ev_end) (void)hipEventRecord(ev_end, st);
  return launch_levels_to_cap(d_tree, n_leaves, cap_height, st, single_proof);
}

// Same as launch_merkle_tree but the leaf digests (level 0) are already in d_tree.
void launch_tree_from_digests(u64* d_tree, size_t n_leaves, unsigned cap_height, hipStream_t st, bool single_proof) {
  launch_levels_to_cap(d_tree, n_leaves, cap_height, st, single_proof);
}

}  // namespace p25
