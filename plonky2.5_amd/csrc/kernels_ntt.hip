// Goldilocks NTT for gfx950: two-pass (four-step) transform with LDS-resident sub-transforms.
//
// Replaces (upstream plonky2_field @ 3de92d9, reached from /root/reference/src/p3/mod.rs:260 via
// PolynomialBatch::from_values / from_coeffs): PolynomialValues::ifft, PolynomialCoeffs::lde +
// coset_fft, and the transpose + reverse_index_bits that orders the LDE as Merkle leaves.
//
// A length-N transform (N = R1*R2) is two launches of ONE kernel, `k_ntt_tile`:
//   pass 1: input index k = k1 + R1*k2.  A block takes a tile of T adjacent k1 (T*8 B contiguous
//           per row -> coalesced) and all R2 values of k2, runs T radix-2 DIF transforms of length
//           R2 in LDS, multiplies by the four-step twiddle w_N^(k1*j2) and stores.
//   pass 2: a block takes T rows (each R1 contiguous words), transforms them in LDS and stores.
// LDS twiddles: the R/2 powers of the sub-transform root are staged in LDS once per block.
// The DIF network leaves its result in bit-reversed LDS order, which is exactly what the prover
// wants: the LDE is stored at bit-reversed index (Merkle-leaf order) with NO separate
// transpose / bit-reversal pass -- pass 1 scatters rows to rev(j2) and pass 2 then runs in place.
// The rate-8 LDE is computed as 8 coset transforms of length n (shift*w_8n^c, c < 8) rather than
// one zero-padded length-8n transform: the same values, 3 fewer butterfly stages, and coset c
// lands in the contiguous leaf block rev3(c).
//
// Addressing kinds (element (t, i): sub-transform t of the launch, element i of it; NT = number
// of sub-transforms per polynomial, R = their length):
//   kind 0 ("t-contiguous"): addr = t + imap(i) * NT
//   kind 1 ("i-contiguous"): addr = tmap(t) * R + imap(i)
// where imap/tmap are the identity or a bit reversal.  All values are canonical field elements.
#include <mutex>
#include <set>
#include <tuple>
#include "kernels.h"
#include "ntt16.h"

// (Phase priorities -- load / store phases of k_ntt_tile above or below its butterflies -- were measured in round 4 and
// change nothing: profiles/r04_ab_ntt_phase_priority.txt.)
// transforms above 2^NTT_FACTOR_LOG points take the coset pre-scale as two factor tables (ntt_lde_bitrev)
constexpr int NTT_FACTOR_LOG = 16;

namespace p25 {

// G consecutive radix-2 DIF stages on the 2^G elements col[k * step], k < 2^G, held in registers.
// Stage s0+m pairs (k, k + hk), hk = 2^(G-1-m); the pair's position inside its half-block is
// j = (k mod hk) * stride + l, twiddle w_R^(j << (s0+m)).  With stride == 1 the last stage has j == 0.
template <int G>
__device__ __forceinline__ void dif_group(u64* col, const u64* wl, int step, int l, int lstride, int s0) {
  constexpr int K = 1 << G;
  u64 x[K];
#pragma unroll
  for (int k = 0; k < K; k++) x[k] = col[k * step];
#pragma unroll
  for (int m = 0; m < G; m++) {
    constexpr int dummy = 0;
    (void)dummy;
    const int hk = K >> (m + 1);
#pragma unroll
    for (int k = 0; k < K; k++) {
      if (k & hk) continue;
      const int j = ((k & (hk - 1)) << lstride) + l;
      u64 sum, d;
      gl::bfly_nc(x[k], x[k + hk], false, sum, d);
      x[k] = sum;
      x[k + hk] = (hk == 1 && lstride == 0) ? d : gl::mul_nc(d, wl[j << (s0 + m)]);
    }
  }
#pragma unroll
  for (int k = 0; k < K; k++) col[k * step] = x[k];
}

// Four DIF stages s0..s0+3 of the length-R sub-transforms as a pure 16-point transform plus ONE twiddle per element:
// the stage-by-stage factors w_R^(((k mod hk) << lstride + l) << (s0 + m)) of the radix-2 network split into the
// 16th-root part (compile-time, above) and w_R^(l << (s0 + m)) on every difference branch, which accumulates to
// w_R^((l * rev4(k')) << s0) at output k' -- 15 general multiplications per 16 elements instead of 32, none in the
// last group of a sub-transform (l = 0).  `wl` holds w_R^e for e < R.
template <bool INV>
__device__ __forceinline__ void dif16_group(u64* col, const u64* wl, int step, int l, int s0) {
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = col[k * step];
  dft16<INV>(x);
  if (l != 0) {
#pragma unroll
    for (int k = 1; k < 16; k++) {
      const int f = ((k & 1) << 3) | ((k & 2) << 1) | ((k & 4) >> 1) | ((k & 8) >> 3);
      x[k] = gl::mul_nc(x[k], wl[(l * f) << s0]);
    }
  }
#pragma unroll
  for (int k = 0; k < 16; k++) col[k * step] = x[k];
}

// PRE: how the input is pre-scaled -- 0 not at all, 1 by one table entry per element (a.pre), 2 by the product of two
// factor-table entries (a.pre_t, a.pre_i).  A template parameter: as run-time branches in the load loop the three forms
// cost the fib-64 shapes 8 % of the kernel (iNTT + LDE of 135 columns 0.95 -> 1.03 ms, round 4).
template <bool INV, int NTH, int PRE>
__global__ __launch_bounds__(NTH) void k_ntt_tile(NttPass a) {
  P25_WAVE_PRIO(P25_PRIO_BULK);
  extern __shared__ u64 lds[];
  const int R = 1 << a.log_r, T = 1 << a.log_t;
  const int TP = T > 1 ? T + 1 : 1;  // row padding: conflict-free for both access directions
  u64* wl = lds + (size_t)R * TP;    // twiddles of the sub-transform: w_R^e, e < R/2 (e < R with full_table)
  const u32 NT = 1u << a.log_nt;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int wstride_log = a.log_n_table - a.log_r;  // w_R^k = w_N^(k * N/R)
  for (int k = tid; k < (a.full_table ? R : R / 2); k += nth) wl[k] = a.pow_table[(size_t)k << wstride_log];
  // Block -> (tile, polynomial, coset), ONE decode for every launch.  The grid is 1-D and workgroups are dealt round-robin
  // to the 8 XCDs, each with its own 4 MB L2: the low log_g bits of the block index (g = min(8, tiles)) pick the tile's
  // residue, so an XCD only ever touches tiles = xcd (mod 8) -- 1/8 of the per-coset pre-scale table, which then stays
  // L2-resident across polynomials -- and the n_cosets blocks reading the SAME coefficient tile run on the same XCD back to
  // back (the tile is fetched from HBM once, not once per coset).  With one coset this is tile = L mod tiles, poly = L / tiles.
  const u32 L = blockIdx.x;
  const u32 m = L >> a.log_g;
  const u32 tiles_g = a.n_tiles >> a.log_g;
  const u32 coset = m % a.n_cosets;
  const u32 r = m / a.n_cosets;
  const u32 tile = (r % tiles_g) * (1u << a.log_g) + (L & ((1u << a.log_g) - 1u));
  const u32 poly = r / tiles_g;
  const u32 tg0 = tile * T;
  const u64* in = a.in + (size_t)poly * a.in_poly_stride;
  u64* out = a.out + (size_t)poly * a.out_poly_stride + a.coset_out_off[coset];
  // coset pre-scale: one table entry per input coefficient (shift_c^k at the coefficient's address k) ...
  const u64* pre = PRE == 1 ? a.pre + ((size_t)coset << (a.log_r + a.log_nt)) : nullptr;
  // ... or its two factors shift_c^t * (shift_c^NT)^i when N is too large for the table to live in L2 (2^19-row circuits:
  // 8 cosets x 4 MB, re-fetched for every polynomial).  A lane's elements all have the same t (the block size is a multiple
  // of the tile width), so the t factor is loaded once; the i factors (R entries per coset) stay cached.
  const u64* pre_i = PRE == 2 ? a.pre_i + ((size_t)coset << a.log_r) : nullptr;
  const u64 pre_tv = PRE == 2 ? a.pre_t[((size_t)coset << a.log_nt) + tg0 + (tid & (T - 1))] : 1;
  auto in_slot = [&](int e, size_t& addr, int& slot, int& ii) {
    int t, i;
    if (a.in_kind == 0) {
      t = e & (T - 1);
      i = e >> a.log_t;
      u32 ipos = a.in_br_i ? gl::bitrev(i, a.log_r) : i;
      addr = (size_t)(tg0 + t) + (size_t)ipos * NT;
    } else {
      int off = e & (R - 1);
      t = e >> a.log_r;
      i = a.in_br_i ? (int)gl::bitrev(off, a.log_r) : off;
      u32 trow = a.in_br_t ? gl::bitrev(tg0 + t, a.log_nt) : tg0 + t;
      addr = (size_t)trow * R + off;
    }
    slot = i * TP + t;
    ii = i;
  };
  // Eight elements per lane at a time: their loads (value + pre-scale factor) are all issued before the first is used.
  // (One element per iteration, as the loop was written first, is one exposed memory round trip per element: the
  // compiler puts s_waitcnt vmcnt(0) right behind each load.)
  constexpr int LB = 8;
  int e = tid;
  for (; e + (LB - 1) * nth < T * R; e += LB * nth) {
    u64 xv[LB], pv[LB];
    int slot[LB];
#pragma unroll
    for (int k = 0; k < LB; k++) {
      size_t addr;
      int ii;
      in_slot(e + k * nth, addr, slot[k], ii);
      xv[k] = in[addr];
      if constexpr (PRE == 1) pv[k] = pre[addr];
      else if constexpr (PRE == 2) pv[k] = pre_i[ii];
      else pv[k] = 1;
    }
#pragma unroll
    for (int k = 0; k < LB; k++) {
      if constexpr (PRE == 1) lds[slot[k]] = gl::mul_nc(xv[k], pv[k]);
      else if constexpr (PRE == 2) lds[slot[k]] = gl::mul_nc(gl::mul_nc(xv[k], pre_tv), pv[k]);
      else lds[slot[k]] = xv[k];
    }
  }
  for (; e < T * R; e += nth) {
    size_t addr;
    int slot, ii;
    in_slot(e, addr, slot, ii);
    u64 x = in[addr];
    if constexpr (PRE == 1) x = gl::mul_nc(x, pre[addr]);
    if constexpr (PRE == 2) x = gl::mul_nc(gl::mul_nc(x, pre_tv), pre_i[ii]);
    lds[slot] = x;
  }
  __syncthreads();

  // DIF network, natural in -> bit-reversed out, up to 4 stages (radix 16) per LDS round trip:
  // a work item holds the 2^g elements {base + k*stride} of one sub-transform in registers.
  for (int s0 = 0; s0 < a.log_r;) {
    const int g = a.log_r - s0 < 4 ? a.log_r - s0 : 4;
    const int lstride = a.log_r - s0 - g;
    const int items = T << (a.log_r - g);
    for (int b = tid; b < items; b += nth) {
      const int t = b & (T - 1), q = b >> a.log_t;
      const int l = q & ((1 << lstride) - 1), h = q >> lstride;
      u64* col = lds + ((h << (a.log_r - s0)) + l) * TP + t;
      switch (g) {
        case 4:
          if (a.full_table)
            dif16_group<INV>(col, wl, TP << lstride, l, s0);
          else
            dif_group<4>(col, wl, TP << lstride, l, lstride, s0);
          break;
        case 3: dif_group<3>(col, wl, TP << lstride, l, lstride, s0); break;
        case 2: dif_group<2>(col, wl, TP << lstride, l, lstride, s0); break;
        default: dif_group<1>(col, wl, TP << lstride, l, lstride, s0); break;
      }
    }
    __syncthreads();
    s0 += g;
  }

  auto out_slot = [&](int e, size_t& addr, int& slot, u32& tw, u32& j) {
    int t, q;
    if (a.out_kind == 0) {
      t = e & (T - 1);
      q = e >> a.log_t;
      j = gl::bitrev(q, a.log_r);   // frequency index held at LDS position q
      u32 ipos = a.out_br_i ? (u32)q : j;
      addr = (size_t)(tg0 + t) + (size_t)ipos * NT;
    } else {
      int off = e & (R - 1);
      t = e >> a.log_r;
      q = a.out_br_i ? off : (int)gl::bitrev(off, a.log_r);
      j = gl::bitrev(q, a.log_r);
      u32 trow = a.out_br_t ? gl::bitrev(tg0 + t, a.log_nt) : tg0 + t;
      addr = (size_t)trow * R + off;
    }
    slot = q * TP + t;
    tw = (u32)(tg0 + t);
  };
  // the four-step twiddles (a gather from the power table) likewise eight at a time
  int eo = tid;
  if (a.use_twiddle && !a.post_t) {
    for (; eo + (LB - 1) * nth < T * R; eo += LB * nth) {
      u64 wv[LB], xv[LB];
      size_t addr[LB];
#pragma unroll
      for (int k = 0; k < LB; k++) {
        int slot;
        u32 tw, j;
        out_slot(eo + k * nth, addr[k], slot, tw, j);
        wv[k] = a.pow_table[(size_t)tw * j];
        xv[k] = lds[slot];
      }
#pragma unroll
      for (int k = 0; k < LB; k++) {
        const u64 y = gl::mul_nc(xv[k], wv[k]);
        out[addr[k]] = a.lazy_out ? y : gl::canon(y);
      }
    }
  }
  for (; eo < T * R; eo += nth) {
    size_t addr;
    int slot;
    u32 tw, j;
    out_slot(eo, addr, slot, tw, j);
    u64 x = lds[slot];   // any u64 (lazy network)
    if (a.use_twiddle) x = gl::mul_nc(x, a.pow_table[(size_t)tw * j]);
    if (a.post_t) x = gl::mul_nc(x, gl::mul_nc(a.post_t[tw], a.post_i[j]));
    out[addr] = a.lazy_out ? x : gl::canon(x);
  }
}

// LDS bytes of a tile: R rows of T (+1 padding) words + the sub-transform's twiddle table
static size_t tile_lds_bytes(int log_r, int log_t) {
  const size_t R = (size_t)1 << log_r, T = (size_t)1 << log_t, TP = T > 1 ? T + 1 : 1;
  return (R * TP + (log_r <= 10 ? R : R / 2)) * sizeof(u64);
}

void launch_ntt_pass(const NttPass& p, int n_polys, int n_cosets, hipStream_t st) {
  NttPass q = p;
  // the radix-16 groups index a full table of R twiddles; a 2^11-point sub-transform (2^22-point transforms) would
  // not fit the LDS a block may take with it and keeps the half table and the radix-2 network
  q.full_table = p.log_r <= 10;
  size_t lds = tile_lds_bytes(p.log_r, p.log_t);
  q.n_tiles = 1u << (p.log_nt - p.log_t);
  q.n_cosets = (uint32_t)n_cosets;
  q.log_g = 0;
  while (q.log_g < 3 && (2u << q.log_g) <= q.n_tiles) q.log_g++;
  const dim3 grid(q.n_tiles * (uint32_t)n_polys * (uint32_t)n_cosets);
  // tiles of 2^13 elements (16-wide tiles of 512-point sub-transforms: 2^19-point transforms, BASELINE config 5) take
  // 512 threads, so that a CU's two resident blocks still give every SIMD four waves; above 64 KB of dynamic LDS the
  // function attribute has to allow it (a workgroup may take up to 160 KB on gfx950)
  const bool wide = p.log_r + p.log_t >= 13;
  const int pre_mode = q.pre ? 1 : (q.pre_t ? 2 : 0);
  auto launch = [&](auto kernel, int nth) {
    if (lds > 64 * 1024) {   // above 64 KB of dynamic LDS the function attribute has to allow it (160 KB per workgroup on gfx950)
      // the attribute belongs to the CURRENT device's function object (the stream pool serves several devices per process)
      static std::mutex mu;
      static std::set<std::pair<int, const void*>> allowed;
      int dev = 0;
      P25_HIP(hipGetDevice(&dev));
      std::lock_guard<std::mutex> lk(mu);
      if (allowed.insert({dev, (const void*)kernel}).second)
        P25_HIP(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(kernel, grid, dim3(nth), lds, st, q);
  };
#define P25_NTT_LAUNCH(INV, NTH)                                                            \
  switch (pre_mode) {                                                                       \
    case 1: launch(k_ntt_tile<INV, NTH, 1>, NTH); break;                                    \
    case 2: launch(k_ntt_tile<INV, NTH, 2>, NTH); break;                                    \
    default: launch(k_ntt_tile<INV, NTH, 0>, NTH); break;                                   \
  }
  if (wide) {
    if (q.inverse) { P25_NTT_LAUNCH(true, 512) } else { P25_NTT_LAUNCH(false, 512) }
  } else {
    if (q.inverse) { P25_NTT_LAUNCH(true, 256) } else { P25_NTT_LAUNCH(false, 256) }
  }
#undef P25_NTT_LAUNCH
}

// Tile width.  A pass whose tile is STRIDED in memory (kind 0: T adjacent words per row of the tile, rows NT words
// apart -- pass 1 of every two-pass transform, reads and writes alike) wants T * 8 B = a whole 128-byte line: the
// widest tile, up to 16, that still lets two blocks share a CU's 160 KB of LDS (R = 256: 37 KB, R = 512: 74 KB; wider
// sub-transforms fall back to 2^12-element tiles: 32-byte segments cost the 2^19-row circuits of BASELINE config 5 an 8x
// over-fetch while their pass 1 was the 1024-point one).  A pass over contiguous rows (kind 1) is coalesced at any
// width and keeps 2^12-element tiles.
static int pick_log_t(int log_r, int log_nt, bool strided) {
  int lt = 12 - log_r;
  if (strided)
    for (int w = 4; w > lt; w--)
      if (tile_lds_bytes(log_r, w) <= 79 * 1024) {
        lt = w;
        break;
      }
  // 1024-point contiguous passes (pass 2 of 2^19- and 2^20-point transforms): 2^12-element tiles take 49 KB, three blocks of
  // four waves per CU; 8-wide tiles of 512 threads take 80 KB, two blocks of eight waves -- four waves per SIMD instead of
  // three to overlap the load, butterfly and store phases.  Config 5: 16.3-16.7 -> 16.8 proofs/s (profiles/r04_ab_config5_ntt.txt).
  if (!strided && log_r == 10) lt = 3;
  if (lt > 4) lt = 4;
  if (lt > log_nt) lt = log_nt;
  if (lt < 0) lt = 0;
  return lt;
}

// values -> coefficients -> LDE for a batch of polynomials: the two-call sequence of the prover's commits.
// (Round 5 measured pass 2 of the inverse transform fused with pass 1 of the LDE in one launch -- the tile the former leaves
// in LDS is the tile the latter loads when n = 2^16 -- as `k_ntt_fused`: bit-exact, 4 % fewer NTT instructions, slower and
// more traffic than the XCD-grouped coset blocks; the kernel lives on as tools/exp/ntt_fused_inverse_pass2_lde_pass1.patch,
// DESIGN section 3.)
void ntt_inverse_then_lde(NttTables& tb, const u64* d_vals, size_t val_stride, u64* d_tmp, size_t tmp_stride, u64* d_coeffs,
                          size_t coeff_stride, u64* d_lde, size_t lde_stride, int log_n, int rate_bits, int n_polys, u64 shift,
                          hipStream_t st) {
  ntt_inverse(tb, d_vals, val_stride, false, d_tmp, tmp_stride, d_coeffs, coeff_stride, log_n, n_polys, 1, st);
  ntt_lde_bitrev(tb, d_coeffs, coeff_stride, d_lde, lde_stride, log_n, rate_bits, n_polys, shift, st);
}

NttTables::~NttTables() {
  for (u64* p : owned_) (void)hipFree(p);
}
const u64* NttTables::upload(const std::vector<u64>& host) {
  u64* d = nullptr;
  P25_HIP(hipMalloc(&d, host.size() * sizeof(u64)));
  owned_.push_back(d);
  P25_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice));
  return d;
}
const u64* NttTables::pow_table(int log_n, bool inverse) {
  auto key = std::make_pair(log_n, inverse);
  auto it = pow_.find(key);
  if (it != pow_.end()) return it->second;
  u64 w = gl::root_of_unity(log_n);
  if (inverse) w = gl::inv(w);
  std::vector<u64> h((size_t)1 << log_n);
  u64 x = 1;
  for (auto& v : h) {
    v = x;
    x = gl::mul(x, w);
  }
  u64* d = const_cast<u64*>(upload(h));
  pow_[key] = d;
  return d;
}
const u64* NttTables::geom_table(u64 first, u64 ratio, size_t len) {
  auto key = std::make_tuple(first, ratio, len);
  auto it = geom_.find(key);
  if (it != geom_.end()) return it->second;
  std::vector<u64> h(len);
  u64 x = first;
  for (auto& v : h) {
    v = x;
    x = gl::mul(x, ratio);
  }
  u64* d = const_cast<u64*>(upload(h));
  geom_[key] = d;
  return d;
}

static void split(int log_n, int& log_r1, int& log_r2) {
  if (log_n <= 10) {
    log_r1 = 0;
    log_r2 = log_n;
  } else {
    log_r2 = log_n / 2;        // pass-1 sub-transform (over k2): the strided pass gets the SMALLER one (wider tiles fit)
    log_r1 = log_n - log_r2;   // pass-2 sub-transform (over k1): contiguous rows
  }
  // sub-transforms of up to 2^11 points (56 KB of LDS with a 2-wide tile): transforms up to 2^22,
  // i.e. the 8n-point quotient iNTT of a 2^19-row circuit (BASELINE config 5)
  if (log_r1 > 11 || log_r2 > 11) throw std::runtime_error("NTT size above 2^22 not supported");
}

void ntt_inverse(NttTables& tb, const u64* d_in, size_t in_stride, bool in_bitrev, u64* d_tmp,
                 size_t tmp_stride, u64* d_out, size_t out_stride, int log_n, int n_polys,
                 u64 coset_shift, hipStream_t st) {
  int l1, l2;
  split(log_n, l1, l2);
  const u64* pw = tb.pow_table(log_n, true);
  u64 n_inv = gl::inv((u64)1 << log_n);
  u64 s_inv = gl::inv(coset_shift);
  NttPass p{};
  p.pow_table = pw;
  p.log_n_table = log_n;
  p.inverse = 1;
  if (l1 == 0) {  // single pass
    p.in = d_in; p.out = d_out;
    p.in_poly_stride = in_stride; p.out_poly_stride = out_stride;
    p.log_r = l2; p.log_nt = 0; p.log_t = 0;
    p.in_kind = 1; p.in_br_i = in_bitrev;
    p.out_kind = 1; p.out_br_i = 0;
    p.post_t = tb.geom_table(n_inv, 1, 1);
    p.post_i = tb.geom_table(1, s_inv, (size_t)1 << l2);
    launch_ntt_pass(p, n_polys, 1, st);
    return;
  }
  // pass 1: t = k1 (NT = R1), i = k2 (R = R2); writes Y[k1][j2] at k1 + R1*j2
  p.in = d_in; p.out = d_tmp;
  p.in_poly_stride = in_stride; p.out_poly_stride = tmp_stride;
  p.log_r = l2; p.log_nt = l1; p.log_t = pick_log_t(l2, l1, true);
  if (in_bitrev) { p.in_kind = 1; p.in_br_t = 1; p.in_br_i = 1; } else { p.in_kind = 0; }
  p.out_kind = 0; p.out_br_i = 0;
  p.use_twiddle = 1;
  p.lazy_out = 1;   // read by pass 2 only
  launch_ntt_pass(p, n_polys, 1, st);
  // pass 2: t = j2 (NT = R2), i = k1 (R = R1); output coefficient j2 + R2*j1, scaled
  NttPass q{};
  q.pow_table = pw; q.log_n_table = log_n;
  q.inverse = 1;
  q.in = d_tmp; q.out = d_out;
  q.in_poly_stride = tmp_stride; q.out_poly_stride = out_stride;
  q.log_r = l1; q.log_nt = l2; q.log_t = pick_log_t(l1, l2, false);
  q.in_kind = 1;
  q.out_kind = 0; q.out_br_i = 0;
  q.post_t = tb.geom_table(n_inv, s_inv, (size_t)1 << l2);
  q.post_i = tb.geom_table(1, gl::pow(s_inv, (u64)1 << l2), (size_t)1 << l1);
  launch_ntt_pass(q, n_polys, 1, st);
}

void ntt_lde_bitrev(NttTables& tb, const u64* d_coeffs, size_t coeff_stride, u64* d_lde,
                    size_t lde_stride, int log_n, int rate_bits, int n_polys, u64 shift,
                    hipStream_t st) {
  int l1, l2;
  split(log_n, l1, l2);
  const int nc = 1 << rate_bits;
  if (nc > 8) throw std::runtime_error("rate_bits > 3 not supported");
  const size_t n = (size_t)1 << log_n;
  const u64* pw = tb.pow_table(log_n, false);
  const u64 w_big = gl::root_of_unity(log_n + rate_bits);
  // per-coset pre-scale table: coefficient k gets shift_c^k, shift_c = shift * w_{8n}^c  (n words per coset;
  // one load + one multiply per element instead of two factor tables and two multiplies) -- while the table (8n words)
  // fits the L2s next to the data (n <= 2^16: 4 MB, 1/8 of it per XCD).  Above that the two-pass transform takes the
  // factors shift_c^k1 (k1 < R1) and (shift_c^R1)^k2 (k2 < R2) instead: 12 K words for n = 2^19 instead of 4 M.
  auto& cache = tb.coset_cache();
  const bool factored = l1 != 0 && log_n > NTT_FACTOR_LOG;
  auto key = std::make_tuple(factored ? -log_n : log_n, rate_bits, shift);
  auto it = cache.find(key);
  if (it == cache.end()) {
    std::vector<u64> h;
    if (!factored) {
      h.resize((size_t)nc * n);
      for (int c = 0; c < nc; c++) {
        u64 sc = gl::mul(shift, gl::pow(w_big, c));
        u64 x = 1;
        for (size_t k = 0; k < n; k++) {
          h[(size_t)c * n + k] = x;
          x = gl::mul(x, sc);
        }
      }
    } else {   // [nc][R1] then [nc][R2]
      const size_t R1 = (size_t)1 << l1, R2 = (size_t)1 << l2;
      h.resize((size_t)nc * (R1 + R2));
      for (int c = 0; c < nc; c++) {
        const u64 sc = gl::mul(shift, gl::pow(w_big, c)), sc_r1 = gl::pow(sc, R1);
        u64 x = 1;
        for (size_t k = 0; k < R1; k++) {
          h[(size_t)c * R1 + k] = x;
          x = gl::mul(x, sc);
        }
        x = 1;
        for (size_t k = 0; k < R2; k++) {
          h[(size_t)nc * R1 + (size_t)c * R2 + k] = x;
          x = gl::mul(x, sc_r1);
        }
      }
    }
    it = cache.emplace(key, tb.upload(h)).first;
  }

  NttPass p{};
  p.pow_table = pw; p.log_n_table = log_n;
  p.in = d_coeffs; p.out = d_lde;
  p.in_poly_stride = coeff_stride; p.out_poly_stride = lde_stride;
  for (int c = 0; c < nc; c++) p.coset_out_off[c] = (size_t)gl::bitrev(c, rate_bits) * n;
  if (factored) {
    p.pre_t = it->second;
    p.pre_i = it->second + ((size_t)nc << l1);
  } else {
    p.pre = it->second;  // indexed by the input address: requires natural-order input (in_br_* = 0)
  }
  if (l1 == 0) {
    p.log_r = l2; p.log_nt = 0; p.log_t = 0;
    p.in_kind = 1; p.out_kind = 1; p.out_br_i = 1;
    launch_ntt_pass(p, n_polys, nc, st);
    return;
  }
  // pass 1: scale, length-R2 transforms over k2, twiddle, store row rev(j2)
  p.log_r = l2; p.log_nt = l1; p.log_t = pick_log_t(l2, l1, true);
  p.in_kind = 0; p.out_kind = 0; p.out_br_i = 1;
  p.use_twiddle = 1;
  p.lazy_out = 1;   // read by pass 2 only (in place)
  launch_ntt_pass(p, n_polys, nc, st);
  // pass 2: every row (all cosets: nc*R2 rows of R1 words) in place, bit-reversed within the row
  NttPass q{};
  q.pow_table = pw; q.log_n_table = log_n;
  q.in = d_lde; q.out = d_lde;
  q.in_poly_stride = lde_stride; q.out_poly_stride = lde_stride;
  q.log_r = l1; q.log_nt = l2 + rate_bits; q.log_t = pick_log_t(l1, l2 + rate_bits, false);
  q.in_kind = 1; q.out_kind = 1; q.out_br_i = 1;
  launch_ntt_pass(q, n_polys, 1, st);
}

}  // namespace p25
