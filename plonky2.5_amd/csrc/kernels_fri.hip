// Openings and FRI kernels.
//
// Replaces upstream plonky2 @ 3de92d9 (reached from /root/reference/src/p3/mod.rs:260):
//   OpeningSet::new                      -> launch_eval_polys   (f(zeta) for every committed polynomial)
//   PolynomialBatch::prove_openings      -> launch_fri_combine  (alpha-batching, division by (X - z), * X)
//   fri_committed_trees                  -> launch_fri_leaf_hash / launch_fri_fold (+ NTT, Merkle)
//   fri_prover_query_rounds              -> launch_queries
// SURVEY.md App. A.7-A.8.  Extension-field vectors are stored as two component arrays (a[], b[]).
#include "kernels.h"
#include "gl_lazy.h"
#include "coop.h"
#include "coop_lat.h"
#include "poseidon.h"
#include "prover_kernels.h"

namespace p25 {

// out[t] = (scale * point)^t for t <= count (count+1 entries), as (a, b) pairs
__global__ void k_ext_pows(const u64* __restrict__ point, u64 scale, uint32_t count, int invert,
                           u64* __restrict__ out) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > count) return;
  gl::E2 z = gl::mul(gl::E2{point[0], point[1]}, scale);
  if (invert) z = gl::inv(z);
  gl::E2 r = gl::e2(1);   // square-and-multiply in lazy arithmetic (gl_lazy.h), canonical at the store
  for (uint32_t e = t; e; e >>= 1) {
    if (e & 1) r = gl::e2_mul_nc(r, z);
    z = gl::e2_mul_nc(z, z);
  }
  out[2 * t] = gl::canon(r.a);
  out[2 * t + 1] = gl::canon(r.b);
}

// one block per polynomial: sum_k c_k z^k with lane t taking the coefficients k = t (mod S).  S = 1024
// lanes: the per-lane Horner chain (n / S dependent extension multiplies) is what a lone proof waits for.
constexpr uint32_t EVAL_LANES = 1024;
constexpr uint32_t EVAL_CHUNK_LOG = 16;  // polynomials longer than 2^16 are split into chunks, one block each
// out[(poly, chunk)] = sum_{k in chunk} c_k z^(k - chunk_start): lane t takes k = t (mod S) within the chunk
__global__ __launch_bounds__(1024) void k_eval_polys(const u64* __restrict__ coeffs, uint32_t log_n, uint32_t log_chunk,
                                                     const u64* __restrict__ pows /*[S+1] ext*/,
                                                     u64* __restrict__ out) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  __shared__ u64 sa[EVAL_LANES], sb[EVAL_LANES];
  const uint32_t n = 1u << log_chunk;
  const uint32_t S = n < EVAL_LANES ? n : EVAL_LANES;
  const uint32_t chunks = 1u << (log_n - log_chunk);
  const u64* c = coeffs + ((size_t)(blockIdx.x / chunks) << log_n) + ((size_t)(blockIdx.x % chunks) << log_chunk);
  const uint32_t t = threadIdx.x;
  gl::E2 acc = gl::e2(0);
  if (t < S) {
    gl::E2 y{pows[2 * S], pows[2 * S + 1]};  // z^S
    for (int k = (int)(n / S) - 1; k >= 0; k--) {   // Horner in lazy arithmetic; the coefficient (canonical) enters with one correction
      acc = gl::e2_mul_nc(acc, y);
      acc.a = gl::add_c(acc.a, c[(size_t)k * S + t]);
    }
    acc = gl::e2_canon(gl::e2_mul_nc(acc, gl::E2{pows[2 * t], pows[2 * t + 1]}));
  }
  sa[t] = acc.a;
  sb[t] = acc.b;
  __syncthreads();
  for (int off = EVAL_LANES / 2; off >= 1; off >>= 1) {
    if (t < (unsigned)off) {
      sa[t] = gl::add(sa[t], sa[t + off]);
      sb[t] = gl::add(sb[t], sb[t + off]);
    }
    __syncthreads();
  }
  if (t == 0) {
    out[2 * blockIdx.x] = sa[0];
    out[2 * blockIdx.x + 1] = sb[0];
  }
}
// out[p] = sum_b part[p][b] (z^chunk)^b  (Horner over the chunks of polynomial p)
__global__ void k_eval_combine(const u64* __restrict__ part, uint32_t n_polys, uint32_t chunks, uint32_t log_chunk,
                               const u64* __restrict__ point, u64 scale, u64* __restrict__ out) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_polys) return;
  gl::E2 y = gl::exp_pow2(gl::mul(gl::E2{point[0], point[1]}, scale), log_chunk);
  gl::E2 acc = gl::e2(0);
  for (int b = (int)chunks - 1; b >= 0; b--) {
    acc = gl::mul(acc, y);
    acc = gl::add(acc, gl::E2{part[2 * ((size_t)p * chunks + b)], part[2 * ((size_t)p * chunks + b) + 1]});
  }
  out[2 * p] = acc.a;
  out[2 * p + 1] = acc.b;
}

void launch_eval_polys(const u64* d_coeffs, uint32_t n_polys, uint32_t log_n, const u64* d_point, u64 scale,
                       u64* d_scratch_pows, u64* d_out, hipStream_t st, bool reuse_pows) {
  const uint32_t log_chunk = log_n < EVAL_CHUNK_LOG ? log_n : EVAL_CHUNK_LOG;
  const uint32_t n = 1u << log_chunk, S = n < EVAL_LANES ? n : EVAL_LANES, chunks = 1u << (log_n - log_chunk);
  if (!reuse_pows)  // (scale * point)^t, t <= S; successive calls at the same point share the table
    hipLaunchKernelGGL(k_ext_pows, dim3((S + 1 + 255) / 256), dim3(256), 0, st, d_point, scale, S, 0, d_scratch_pows);
  if (chunks == 1) {
    hipLaunchKernelGGL(k_eval_polys, dim3(n_polys), dim3(EVAL_LANES), 0, st, d_coeffs, log_n, log_chunk, d_scratch_pows, d_out);
    return;
  }
  // partial sums behind the power table: [n_polys][chunks] extension values
  u64* part = d_scratch_pows + 2 * (EVAL_LANES + 2);
  hipLaunchKernelGGL(k_eval_polys, dim3(n_polys * chunks), dim3(EVAL_LANES), 0, st, d_coeffs, log_n, log_chunk, d_scratch_pows, part);
  hipLaunchKernelGGL(k_eval_combine, dim3((n_polys + 63) / 64), dim3(64), 0, st, part, n_polys, chunks, log_chunk, d_point, scale, d_out);
}

// ---------------------------------------------------------------- FRI batching
// comp layout: [batch][component][n]; written already multiplied by z_b^k (d_k = c_k z^k)
struct CombineK {
  const u64* coeffs[4];
  uint32_t n_polys[4];
  uint32_t log_n, num_challenges;
  const u64* alpha_pows;  // ext alpha^j, j <= total polys
  const u64* zpow[2];     // ext z_b^k, k <= n
  u64* comp;
};
__global__ __launch_bounds__(256) void k_fri_comp(CombineK a) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  extern __shared__ u64 ap[];  // alpha^j as (a,b), j < total
  uint32_t total = a.n_polys[0] + a.n_polys[1] + a.n_polys[2] + a.n_polys[3];
  for (uint32_t i = threadIdx.x; i < 2 * total; i += blockDim.x) ap[i] = a.alpha_pows[i];
  __syncthreads();
  const uint32_t n = 1u << a.log_n;
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  gl::E2 s0 = gl::e2(0);
  uint32_t j = 0;
  for (int o = 0; o < 4; o++)
    for (uint32_t p = 0; p < a.n_polys[o]; p++, j++) {
      u64 c = a.coeffs[o][(size_t)p * n + k];
      s0.a = gl::mad_nc(ap[2 * j], c, s0.a);       // any-u64 running sums, canonical at the store
      s0.b = gl::mad_nc(ap[2 * j + 1], c, s0.b);
    }
  gl::E2 s1 = gl::e2(0);
  for (uint32_t c = 0; c < a.num_challenges; c++) {
    u64 v = a.coeffs[2][(size_t)c * n + k];
    s1.a = gl::mad_nc(ap[2 * c], v, s1.a);
    s1.b = gl::mad_nc(ap[2 * c + 1], v, s1.b);
  }
  gl::E2 d0 = gl::e2_canon(gl::e2_mul_nc(s0, gl::E2{a.zpow[0][2 * k], a.zpow[0][2 * k + 1]}));
  gl::E2 d1 = gl::e2_canon(gl::e2_mul_nc(s1, gl::E2{a.zpow[1][2 * k], a.zpow[1][2 * k + 1]}));
  a.comp[0 * (size_t)n + k] = d0.a;
  a.comp[1 * (size_t)n + k] = d0.b;
  a.comp[2 * (size_t)n + k] = d1.a;
  a.comp[3 * (size_t)n + k] = d1.b;
}

// exclusive SUFFIX sums (field addition) of `count` arrays of length n, in place:
// data[k] <- sum_{j > k} data[j].  Three phases; index reversed so that it is a prefix scan.
__global__ __launch_bounds__(256) void k_sfx_block(u64* data, uint32_t n, u64* block_tot) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 sh[256];
  u64* d = data + (size_t)blockIdx.y * n;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;  // reversed index
  u64 v = i < n ? d[n - 1 - i] : 0;
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    u64 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    v = gl::add(v, t);
    sh[threadIdx.x] = v;
    __syncthreads();
  }
  // exclusive within block
  u64 ex = threadIdx.x == 0 ? 0 : sh[threadIdx.x - 1];
  if (i < n) d[n - 1 - i] = ex;
  if (threadIdx.x == 255) block_tot[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_sfx_totals(u64* block_tot, uint32_t n_blocks) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 sh[256];
  __shared__ u64 carry_s;
  u64* bt = block_tot + (size_t)blockIdx.x * n_blocks;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 256) {
    uint32_t i = base + threadIdx.x;
    u64 v = i < n_blocks ? bt[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
      u64 t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
      __syncthreads();
      v = gl::add(v, t);
      sh[threadIdx.x] = v;
      __syncthreads();
    }
    u64 carry = carry_s;
    u64 excl = threadIdx.x == 0 ? carry : gl::add(carry, sh[threadIdx.x - 1]);
    __syncthreads();
    if (i < n_blocks) bt[i] = excl;
    if (threadIdx.x == 255) carry_s = gl::add(carry, v);
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_sfx_apply(u64* data, uint32_t n, const u64* block_tot) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  u64* d = data + (size_t)blockIdx.y * n;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  d[n - 1 - i] = gl::add(d[n - 1 - i], block_tot[(size_t)blockIdx.y * gridDim.x + blockIdx.x]);
}

// final[0] = 0; final[k+1] = alpha^NC * q0[k] + q1[k],  q_b[k] = S_b[k] * z_b^-(k+1)
__global__ __launch_bounds__(256) void k_fri_final(const u64* __restrict__ S, uint32_t n,
                                                   const u64* __restrict__ zinv0, const u64* __restrict__ zinv1,
                                                   const u64* __restrict__ alpha_pows, uint32_t nc,
                                                   u64* __restrict__ fa, u64* __restrict__ fb) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  if (k == n - 1) {
    fa[0] = 0;
    fb[0] = 0;
    return;
  }
  gl::E2 q0 = gl::mul(gl::E2{S[k], S[(size_t)n + k]}, gl::E2{zinv0[2 * (k + 1)], zinv0[2 * (k + 1) + 1]});
  gl::E2 q1 = gl::mul(gl::E2{S[2 * (size_t)n + k], S[3 * (size_t)n + k]}, gl::E2{zinv1[2 * (k + 1)], zinv1[2 * (k + 1) + 1]});
  gl::E2 f = gl::add(gl::mul(q0, gl::E2{alpha_pows[2 * nc], alpha_pows[2 * nc + 1]}), q1);
  fa[k + 1] = f.a;
  fb[k + 1] = f.b;
}

void launch_fri_combine(const FriCombineArgs& a, hipStream_t st) {
  const uint32_t n = 1u << a.log_n;
  uint32_t total = a.n_polys[0] + a.n_polys[1] + a.n_polys[2] + a.n_polys[3];
  // scratch carve-up inside scan_tmp: alpha pows [2*(total+1)], zpow0, zpow1, zinv0, zinv1 [2*(n+1)] each, block totals
  u64* alpha_pows = a.scan_tmp;
  u64* zpow0 = alpha_pows + 2 * (size_t)(total + 1);
  u64* zpow1 = zpow0 + 2 * (size_t)(n + 1);
  u64* zinv0 = zpow1 + 2 * (size_t)(n + 1);
  u64* zinv1 = zinv0 + 2 * (size_t)(n + 1);
  u64* block_tot = zinv1 + 2 * (size_t)(n + 1);
  const unsigned nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_ext_pows, dim3((total + 1 + 255) / 256), dim3(256), 0, st, a.chal + CH_FRI_ALPHA, (u64)1, total, 0, alpha_pows);
  hipLaunchKernelGGL(k_ext_pows, dim3((n + 1 + 255) / 256), dim3(256), 0, st, a.chal + CH_ZETA, (u64)1, n, 0, zpow0);
  hipLaunchKernelGGL(k_ext_pows, dim3((n + 1 + 255) / 256), dim3(256), 0, st, a.chal + CH_ZETA, a.g, n, 0, zpow1);
  hipLaunchKernelGGL(k_ext_pows, dim3((n + 1 + 255) / 256), dim3(256), 0, st, a.chal + CH_ZETA, (u64)1, n, 1, zinv0);
  hipLaunchKernelGGL(k_ext_pows, dim3((n + 1 + 255) / 256), dim3(256), 0, st, a.chal + CH_ZETA, a.g, n, 1, zinv1);
  CombineK ck;
  for (int o = 0; o < 4; o++) {
    ck.coeffs[o] = a.coeffs[o];
    ck.n_polys[o] = a.n_polys[o];
  }
  ck.log_n = a.log_n;
  ck.num_challenges = a.num_challenges;
  ck.alpha_pows = alpha_pows;
  ck.zpow[0] = zpow0;
  ck.zpow[1] = zpow1;
  ck.comp = a.comp;
  hipLaunchKernelGGL(k_fri_comp, dim3(nb), dim3(256), 2 * total * sizeof(u64), st, ck);
  hipLaunchKernelGGL(k_sfx_block, dim3(nb, 4), dim3(256), 0, st, a.comp, n, block_tot);
  hipLaunchKernelGGL(k_sfx_totals, dim3(4), dim3(256), 0, st, block_tot, nb);
  hipLaunchKernelGGL(k_sfx_apply, dim3(nb, 4), dim3(256), 0, st, a.comp, n, block_tot);
  hipLaunchKernelGGL(k_fri_final, dim3(nb), dim3(256), 0, st, a.comp, n, zinv0, zinv1, alpha_pows, a.num_challenges, a.final_a, a.final_b);
}

// ---------------------------------------------------------------- FRI layers
__global__ __launch_bounds__(256) void k_fri_fold(const u64* __restrict__ ca, const u64* __restrict__ cb,
                                                  uint32_t len_out, uint32_t arity_bits,
                                                  const u64* __restrict__ beta, u64* __restrict__ oa,
                                                  u64* __restrict__ ob) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len_out) return;
  const uint32_t arity = 1u << arity_bits;
  gl::E2 b{beta[0], beta[1]};
  gl::E2 s = gl::e2(0);
  for (int k = (int)arity - 1; k >= 0; k--) {
    s = gl::mul(s, b);
    s = gl::add(s, gl::E2{ca[(size_t)i * arity + k], cb[(size_t)i * arity + k]});
  }
  oa[i] = s.a;
  ob[i] = s.b;
}
void launch_fri_fold(const u64* ca, const u64* cb, uint32_t len_out, uint32_t arity_bits, const u64* d_beta,
                     u64* oa, u64* ob, hipStream_t st) {
  hipLaunchKernelGGL(k_fri_fold, dim3((len_out + 255) / 256), dim3(256), 0, st, ca, cb, len_out, arity_bits, d_beta, oa, ob);
}

__global__ __launch_bounds__(256) void k_fri_leaf_hash(const u64* __restrict__ va, const u64* __restrict__ vb,
                                                       uint32_t n_leaves, uint32_t arity_bits,
                                                       u64* __restrict__ digests) {
  uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_leaves) return;
  const uint32_t arity = 1u << arity_bits, words = 2 * arity;
  u64 s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = 0;
  u64 out[4];
  if (words <= 4) {
    for (uint32_t i = 0; i < 4; i++) out[i] = i < words ? ((i & 1) ? vb : va)[(size_t)l * arity + (i >> 1)] : 0;
  } else {
    for (uint32_t off = 0; off < words; off += 8) {
      uint32_t m = words - off < 8 ? words - off : 8;
      for (uint32_t i = 0; i < m; i++) {
        uint32_t wd = off + i;
        s[i] = ((wd & 1) ? vb : va)[(size_t)l * arity + (wd >> 1)];
      }
      poseidon::permute(s);
    }
    for (int i = 0; i < 4; i++) out[i] = s[i];
  }
  for (int i = 0; i < 4; i++) digests[4 * (size_t)l + i] = out[i];
}
// Same digests, one 16-lane group per leaf (coop.h).  A FRI layer has few leaves (2^15, 2^11, 2^7 for the
// fib-64 circuit) and each needs 4 chained permutations, so the per-lane kernel is pure latency (~250 us per
// layer whatever its size); cooperatively the chain is ~4 x 12 us.  Used when a single proof is in flight.
__global__ __launch_bounds__(256) void k_fri_leaf_hash_coop(const u64* __restrict__ va, const u64* __restrict__ vb,
                                                            uint32_t n_leaves, uint32_t arity_bits,
                                                            u64* __restrict__ digests) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  __shared__ u64 rc_lds[360];
  coop::stage_poseidon_rc(rc_lds);
  size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / coop::GROUP;
  const int rr = threadIdx.x & (coop::GROUP - 1);
  const bool valid = g < n_leaves;
  if (!valid) g = n_leaves - 1;  // keep every lane in the shuffles
  const uint32_t arity = 1u << arity_bits, words = 2 * arity;
  u64 s = 0;
  for (uint32_t off = 0; off < words; off += 8) {
    const uint32_t m = words - off < 8 ? words - off : 8;
    const uint32_t wd = off + (uint32_t)rr;
    if ((uint32_t)rr < m) s = ((wd & 1) ? vb : va)[g * arity + (wd >> 1)];
    s = coop::poseidon_permute_lat(s, threadIdx.x & 63, rc_lds);
  }
  if (valid && rr < 4) digests[4 * g + rr] = s;
}
void launch_fri_leaf_hash(const u64* va, const u64* vb, uint32_t n_leaves, uint32_t arity_bits, u64* d_digests,
                          hipStream_t st, bool single_proof) {
  // cooperative form only when a lone proof is in flight: it costs 4x the instructions of the per-lane form
  if ((2u << arity_bits) <= 4 || !single_proof) {
    hipLaunchKernelGGL(k_fri_leaf_hash, dim3((n_leaves + 255) / 256), dim3(256), 0, st, va, vb, n_leaves, arity_bits, d_digests);
    return;
  }
  const size_t th = (size_t)n_leaves * coop::GROUP;
  hipLaunchKernelGGL(k_fri_leaf_hash_coop, dim3((unsigned)((th + 255) / 256)), dim3(256), 0, st, va, vb, n_leaves,
                     arity_bits, d_digests);
}

// ---------------------------------------------------------------- queries
__device__ __forceinline__ size_t level_off(size_t n_leaves, uint32_t k) { return 8 * n_leaves - ((8 * n_leaves) >> k); }

__global__ __launch_bounds__(256) void k_queries(QueryArgs a) {
  P25_WAVE_PRIO(P25_PRIO_CHAIN);
  const uint32_t q = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const size_t big = (size_t)1 << a.lde_bits;
  size_t x = (size_t)(a.chal[CH_QUERIES + q] & (big - 1));  // challenge mod 2^lde_bits
  u64* out = a.proof + a.query_offset + (size_t)q * a.query_stride;
  const uint32_t path_len = a.lde_bits - a.cap_height;
  for (uint32_t o = 0; o < a.n_oracles; o++) {
    for (uint32_t c = t; c < a.oracle_width[o]; c += nt) out[c] = a.oracle_lde[o][(size_t)c * big + x];
    out += a.oracle_width[o];
    for (uint32_t e = t; e < 4 * path_len; e += nt) {
      uint32_t k = e >> 2;
      out[e] = a.oracle_tree[o][level_off(big, k) + 4 * ((x >> k) ^ 1) + (e & 3)];
    }
    out += 4 * path_len;
  }
  uint32_t bits = a.lde_bits;
  for (uint32_t l = 0; l < a.n_layers; l++) {
    const uint32_t ab = a.arity_bits[l], arity = 1u << ab;
    bits -= ab;
    const size_t n_leaves = (size_t)1 << bits;
    const size_t ci = x >> ab;
    for (uint32_t e = t; e < 2 * arity; e += nt) out[e] = ((e & 1) ? a.layer_vb[l] : a.layer_va[l])[ci * arity + (e >> 1)];
    out += 2 * arity;
    const uint32_t pl = bits - a.cap_height;
    for (uint32_t e = t; e < 4 * pl; e += nt) {
      uint32_t k = e >> 2;
      out[e] = a.layer_tree[l][level_off(n_leaves, k) + 4 * ((ci >> k) ^ 1) + (e & 3)];
    }
    out += 4 * pl;
    x = ci;
  }
}
void launch_queries(const QueryArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_queries, dim3(a.num_queries), dim3(256), 0, st, a);
}

}  // namespace p25
