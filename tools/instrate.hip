// Measures issue rates of individual gfx950 VALU instructions (inline asm, independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define REP8(x) x x x x x x x x
#define KERNEL(NAME, ASM, CONSTR...)                                                     \
  __global__ __launch_bounds__(256) void NAME(uint64_t* out, int iters) {               \
    uint32_t a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u, c = a + 77, d = b * 3 + 1; \
    uint64_t x = a, y = b, z = c, w = d;                                                \
    for (int i = 0; i < iters; i++) { REP8(asm volatile(ASM : CONSTR);) }              \
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + x + y + z + w;               \
  }
KERNEL(k_add_u32, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x))
KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x))
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x | 1))
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x | 0x80000001u))
KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x | 1))
KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x | 1))
KERNEL(k_mad64, "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3", "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b) : "vcc")
KERNEL(k_lshl_add64, "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n v_lshl_add_u64 %3, %3, 0, %0", "+v"(x), "+v"(y), "+v"(z), "+v"(w))
KERNEL(k_add_co, "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x) : "vcc")
KERNEL(k_cndmask, "v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x) : "vcc")
KERNEL(k_fma64, "v_fma_f64 %0, %0, %0, %1\n v_fma_f64 %1, %1, %1, %2\n v_fma_f64 %2, %2, %2, %3\n v_fma_f64 %3, %3, %3, %0", "+v"(x), "+v"(y), "+v"(z), "+v"(w))
KERNEL(k_mul_u64lo, "v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x | 0x80000001u))
KERNEL(k_dot4, "v_dot4_u32_u8 %0, %0, %4, %1\n v_dot4_u32_u8 %1, %1, %4, %2\n v_dot4_u32_u8 %2, %2, %4, %3\n v_dot4_u32_u8 %3, %3, %4, %0", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x | 0x01010101u))
KERNEL(k_mad_i32_i16, "v_mad_u32_u16 %0, %0, %4, %1\n v_mad_u32_u16 %1, %1, %4, %2\n v_mad_u32_u16 %2, %2, %4, %3\n v_mad_u32_u16 %3, %3, %4, %0", "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x | 1))

// ---- round 6: the CARRY forms priced one by one at the residency of the hash kernels (6 waves per SIMD) -------------
// The proving pipeline's VALU ceiling ("one wave-instruction per SIMD per 4 cycles") rests on the instruction mix of the Poseidon
// sponge: v_mad_u64_u32 plus carry / borrow ops in two encodings -- VOP2 `_e32` (carry in VCC) and VOP3 `_e64` (carry in an SGPR
// pair).  Each form below runs alone: 64-lane workgroups, exactly 24 of them per CU (= 6 waves per SIMD: 6,800 B of LDS per
// workgroup make 24 fit and 25 not), one full residency round (256 CUs x 24), eight independent chains per wave, the shader clock
// measured in the same kernel (cycle counter vs the constant-rate wall clock).  Values are irrelevant; only issue slots are counted.
#define REP4(x) x x x x
#define CARRY_KERNEL(NAME, ASM8)                                                                                \
  __global__ __launch_bounds__(256) void NAME(uint64_t* out, int iters, unsigned long long* clk) {             \
    extern __shared__ uint64_t pad_[];                                                                          \
    uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 * 3 + 1;          \
    uint32_t a4 = a0 * 5 + 3, a5 = a1 ^ 0x7f4a7c15u, a6 = a2 * 7 + 5, a7 = a3 ^ 0x85ebca6bu;                     \
    uint64_t x0 = a0 | ((uint64_t)a1 << 32), x1 = a2 | ((uint64_t)a3 << 32), x2 = a4 | ((uint64_t)a5 << 32),   \
             x3 = a6 | ((uint64_t)a7 << 32);                                                                     \
    const uint32_t k = threadIdx.x | 0x80000001u;                                                               \
    uint64_t s0, s1;                                                                                            \
    asm volatile("s_mov_b64 %0, 0x5555\n s_mov_b64 %1, 0x3333" : "=s"(s0), "=s"(s1));                         \
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();                            \
    for (int i = 0; i < iters; i++) {                                                                           \
      REP4(asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), \
                               "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+s"(s0), "+s"(s1) : "v"(k) : "vcc");)   \
    }                                                                                                           \
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();                            \
    if (threadIdx.x == 0 && (blockIdx.x & 255) == 0) { clk[2 * (blockIdx.x >> 8)] = c1 - c0; clk[2 * (blockIdx.x >> 8) + 1] = w1 - w0; } \
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + x0 + x1 + x2 + x3 + s0 + s1 + (pad_[0] & 0); \
  }
// operands: %0..%7 32-bit VGPRs, %8..%11 64-bit VGPR pairs, %12 / %13 SGPR pairs, %14 a 32-bit VGPR input
CARRY_KERNEL(c_add_u32_e32,   "v_add_u32_e32 %0, %0, %14\n v_add_u32_e32 %1, %1, %14\n v_add_u32_e32 %2, %2, %14\n v_add_u32_e32 %3, %3, %14\n v_add_u32_e32 %4, %4, %14\n v_add_u32_e32 %5, %5, %14\n v_add_u32_e32 %6, %6, %14\n v_add_u32_e32 %7, %7, %14")
CARRY_KERNEL(c_add_co_e32,    "v_add_co_u32_e32 %0, vcc, %0, %14\n v_add_co_u32_e32 %1, vcc, %1, %14\n v_add_co_u32_e32 %2, vcc, %2, %14\n v_add_co_u32_e32 %3, vcc, %3, %14\n v_add_co_u32_e32 %4, vcc, %4, %14\n v_add_co_u32_e32 %5, vcc, %5, %14\n v_add_co_u32_e32 %6, vcc, %6, %14\n v_add_co_u32_e32 %7, vcc, %7, %14")
CARRY_KERNEL(c_addc_co_e32,   "v_addc_co_u32_e32 %0, vcc, %0, %14, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %14, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %14, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %14, vcc\n v_addc_co_u32_e32 %4, vcc, %4, %14, vcc\n v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n v_addc_co_u32_e32 %6, vcc, %6, %14, vcc\n v_addc_co_u32_e32 %7, vcc, %7, %14, vcc")
CARRY_KERNEL(c_add_co_e64,    "v_add_co_u32_e64 %0, %12, %0, %14\n v_add_co_u32_e64 %1, %12, %1, %14\n v_add_co_u32_e64 %2, %12, %2, %14\n v_add_co_u32_e64 %3, %12, %3, %14\n v_add_co_u32_e64 %4, %12, %4, %14\n v_add_co_u32_e64 %5, %12, %5, %14\n v_add_co_u32_e64 %6, %12, %6, %14\n v_add_co_u32_e64 %7, %12, %7, %14")
// carry in from one SGPR pair (written by SALU before the loop), carry out to another: no VALU-written SGPR is read by a VALU
CARRY_KERNEL(c_addc_co_e64,   "v_addc_co_u32_e64 %0, %12, %0, %14, %13\n v_addc_co_u32_e64 %1, %12, %1, %14, %13\n v_addc_co_u32_e64 %2, %12, %2, %14, %13\n v_addc_co_u32_e64 %3, %12, %3, %14, %13\n v_addc_co_u32_e64 %4, %12, %4, %14, %13\n v_addc_co_u32_e64 %5, %12, %5, %14, %13\n v_addc_co_u32_e64 %6, %12, %6, %14, %13\n v_addc_co_u32_e64 %7, %12, %7, %14, %13")
CARRY_KERNEL(c_sub_co_e32,    "v_sub_co_u32_e32 %0, vcc, %0, %14\n v_sub_co_u32_e32 %1, vcc, %1, %14\n v_sub_co_u32_e32 %2, vcc, %2, %14\n v_sub_co_u32_e32 %3, vcc, %3, %14\n v_sub_co_u32_e32 %4, vcc, %4, %14\n v_sub_co_u32_e32 %5, vcc, %5, %14\n v_sub_co_u32_e32 %6, vcc, %6, %14\n v_sub_co_u32_e32 %7, vcc, %7, %14")
CARRY_KERNEL(c_subb_co_e64,   "v_subb_co_u32_e64 %0, %12, %0, %14, %13\n v_subb_co_u32_e64 %1, %12, %1, %14, %13\n v_subb_co_u32_e64 %2, %12, %2, %14, %13\n v_subb_co_u32_e64 %3, %12, %3, %14, %13\n v_subb_co_u32_e64 %4, %12, %4, %14, %13\n v_subb_co_u32_e64 %5, %12, %5, %14, %13\n v_subb_co_u32_e64 %6, %12, %6, %14, %13\n v_subb_co_u32_e64 %7, %12, %7, %14, %13")
// the dependent 64-bit add as the compiler writes it: add_co then addc through VCC (4 pairs)
CARRY_KERNEL(c_pair_e32,      "v_add_co_u32_e32 %0, vcc, %0, %14\n v_addc_co_u32_e32 %1, vcc, %1, %14, vcc\n v_add_co_u32_e32 %2, vcc, %2, %14\n v_addc_co_u32_e32 %3, vcc, %3, %14, vcc\n v_add_co_u32_e32 %4, vcc, %4, %14\n v_addc_co_u32_e32 %5, vcc, %5, %14, vcc\n v_add_co_u32_e32 %6, vcc, %6, %14\n v_addc_co_u32_e32 %7, vcc, %7, %14, vcc")
CARRY_KERNEL(c_lshl_add_u64,  "v_lshl_add_u64 %8, %8, 0, %9\n v_lshl_add_u64 %9, %9, 0, %10\n v_lshl_add_u64 %10, %10, 0, %11\n v_lshl_add_u64 %11, %11, 0, %8\n v_lshl_add_u64 %8, %8, 0, %10\n v_lshl_add_u64 %9, %9, 0, %11\n v_lshl_add_u64 %10, %10, 0, %8\n v_lshl_add_u64 %11, %11, 0, %9")
// v_mad_u64_u32 as a 64-bit add-with-carry: acc + x * 1, carry out to VCC / to an SGPR pair
CARRY_KERNEL(c_mad64_x1_vcc,  "v_mad_u64_u32 %8, vcc, %0, 1, %8\n v_mad_u64_u32 %9, vcc, %1, 1, %9\n v_mad_u64_u32 %10, vcc, %2, 1, %10\n v_mad_u64_u32 %11, vcc, %3, 1, %11\n v_mad_u64_u32 %8, vcc, %4, 1, %8\n v_mad_u64_u32 %9, vcc, %5, 1, %9\n v_mad_u64_u32 %10, vcc, %6, 1, %10\n v_mad_u64_u32 %11, vcc, %7, 1, %11")
CARRY_KERNEL(c_mad64_x1_sgpr, "v_mad_u64_u32 %8, %12, %0, 1, %8\n v_mad_u64_u32 %9, %12, %1, 1, %9\n v_mad_u64_u32 %10, %12, %2, 1, %10\n v_mad_u64_u32 %11, %12, %3, 1, %11\n v_mad_u64_u32 %8, %12, %4, 1, %8\n v_mad_u64_u32 %9, %12, %5, 1, %9\n v_mad_u64_u32 %10, %12, %6, 1, %10\n v_mad_u64_u32 %11, %12, %7, 1, %11")
// the full multiply-add (both factors variable), for the same residency
CARRY_KERNEL(c_mad64_full,    "v_mad_u64_u32 %8, vcc, %0, %14, %8\n v_mad_u64_u32 %9, vcc, %1, %14, %9\n v_mad_u64_u32 %10, vcc, %2, %14, %10\n v_mad_u64_u32 %11, vcc, %3, %14, %11\n v_mad_u64_u32 %8, vcc, %4, %14, %8\n v_mad_u64_u32 %9, vcc, %5, %14, %9\n v_mad_u64_u32 %10, vcc, %6, %14, %10\n v_mad_u64_u32 %11, vcc, %7, %14, %11")
// ... as the Poseidon MDS layers issue it: the second factor an inline constant, or an SGPR (the M^3 rows of the partial rounds)
CARRY_KERNEL(c_mad64_inline,  "v_mad_u64_u32 %8, vcc, %0, 17, %8\n v_mad_u64_u32 %9, vcc, %1, 41, %9\n v_mad_u64_u32 %10, vcc, %2, 16, %10\n v_mad_u64_u32 %11, vcc, %3, 28, %11\n v_mad_u64_u32 %8, vcc, %4, 13, %8\n v_mad_u64_u32 %9, vcc, %5, 39, %9\n v_mad_u64_u32 %10, vcc, %6, 18, %10\n v_mad_u64_u32 %11, vcc, %7, 34, %11")
CARRY_KERNEL(c_mad64_sgpr_op, "v_mad_u64_u32 %8, vcc, %0, s20, %8\n v_mad_u64_u32 %9, vcc, %1, s21, %9\n v_mad_u64_u32 %10, vcc, %2, s22, %10\n v_mad_u64_u32 %11, vcc, %3, s23, %11\n v_mad_u64_u32 %8, vcc, %4, s20, %8\n v_mad_u64_u32 %9, vcc, %5, s21, %9\n v_mad_u64_u32 %10, vcc, %6, s22, %10\n v_mad_u64_u32 %11, vcc, %7, s23, %11")
CARRY_KERNEL(c_cndmask_e64,   "v_cndmask_b32_e64 %0, %0, %14, %13\n v_cndmask_b32_e64 %1, %1, %14, %13\n v_cndmask_b32_e64 %2, %2, %14, %13\n v_cndmask_b32_e64 %3, %3, %14, %13\n v_cndmask_b32_e64 %4, %4, %14, %13\n v_cndmask_b32_e64 %5, %5, %14, %13\n v_cndmask_b32_e64 %6, %6, %14, %13\n v_cndmask_b32_e64 %7, %7, %14, %13")
CARRY_KERNEL(c_cmp_e64,       "v_cmp_lt_u32_e64 %12, %0, %14\n v_cmp_lt_u32_e64 %12, %1, %14\n v_cmp_lt_u32_e64 %12, %2, %14\n v_cmp_lt_u32_e64 %12, %3, %14\n v_cmp_lt_u32_e64 %12, %4, %14\n v_cmp_lt_u32_e64 %12, %5, %14\n v_cmp_lt_u32_e64 %12, %6, %14\n v_cmp_lt_u32_e64 %12, %7, %14")

// wg: lanes per workgroup; per_simd: waves per SIMD the launch is held to (by LDS: a CU has 160 KB); one residency round exactly
template <class K> double run6(const char* name, K k, double* clock_hz_out, int wg = 64, int per_simd = 6) {
  int dev = 0, cus = 0, khz = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
  const int wg_per_cu = per_simd * 4 * 64 / wg, blocks = cus * wg_per_cu, iters = 20000, per_iter = 32;   // 4 asm blocks of 8 per iteration
  const size_t lds = per_simd >= 8 ? 0 : (size_t)(160 * 1024 / wg_per_cu) / 256 * 256 - 256;   // wg_per_cu fit, one more does not
  const int probes = (blocks + 255) / 256;
  uint64_t* d; (void)hipMalloc(&d, (size_t)blocks * wg * 8 + 16 * probes);
  unsigned long long* clk = (unsigned long long*)(d + (size_t)blocks * wg);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(wg), lds, 0, d, 200, clk);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(wg), lds, 0, d, iters, clk);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long* h = (unsigned long long*)malloc(16 * probes);
  (void)hipMemcpy(h, clk, 16 * probes, hipMemcpyDeviceToHost);
  // shader clock: the probe waves' cycle counts over their wall-clock ticks, summed (the oldest wave of a SIMD is served first, so
  // the probes finish at different times: together they sample the whole launch); longest probe = how long a wave lives
  double cyc_sum = 0, tick_sum = 0, longest = 0;
  for (int i = 0; i < probes; i++) { cyc_sum += (double)h[2 * i]; tick_sum += (double)h[2 * i + 1]; if ((double)h[2 * i + 1] > longest) longest = (double)h[2 * i + 1]; }
  const double hz = cyc_sum / (tick_sum / ((double)khz * 1e3));
  const double wave_instr = (double)blocks * (wg / 64) * iters * per_iter;
  const double cyc = ms * 1e-3 * hz * (cus * 4) / wave_instr;       // cycles per wave64 instruction per SIMD
  printf("%-34s wg %3d x %2d/SIMD %8.3f ms  clock %.3f GHz  %5.2f cycles per wave64 instruction per SIMD  (longest-lived probe wave: %3.0f %% of the launch)\n",
         name, wg, per_simd, ms, hz / 1e9, cyc, 100.0 * (longest / ((double)khz * 1e3)) / (ms * 1e-3));
  if (clock_hz_out) *clock_hz_out = hz;
  free(h);
  (void)hipFree(d);
  return cyc;
}
template <class K> void run(const char* name, K k, int per_iter) {
  const int blocks = 256 * 16, iters = 4000;
  uint64_t* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double total = (double)blocks * 256 * iters * 8 * per_iter;
  printf("%-16s %8.3f ms  %6.1f lane-instr/clk/CU  (%.2f cycles per wave64 instr per SIMD)\n", name, ms,
         total / (ms * 1e-3) / 256 / 2.4e9, 64.0 / (total / (ms * 1e-3) / 256 / 2.4e9 / 4));
  (void)hipFree(d);
}
int main() {
  run("v_add_u32", k_add_u32, 4); run("v_add3_u32", k_add3_u32, 4); run("v_mul_lo_u32", k_mul_lo, 4);
  run("v_mul_hi_u32", k_mul_hi, 4); run("v_mul_u32_u24", k_mul24, 4); run("v_mad_u32_u24", k_mad24, 4);
  run("v_mad_u64_u32", k_mad64, 4); run("v_lshl_add_u64", k_lshl_add64, 4); run("add_co+addc", k_add_co, 4);
  run("cmp+cndmask", k_cndmask, 4); run("v_fma_f64", k_fma64, 4); run("mul_lo+mul_hi", k_mul_u64lo, 4);
  run("v_dot4_u32_u8", k_dot4, 4); run("v_mad_u32_u16", k_mad_i32_i16, 4);
  printf("\n-- carry forms one by one, 6 waves per SIMD resident (24 x 64-lane workgroups per CU), measured clock --\n");
  double hz = 0;
  const double add32 = run6("v_add_u32_e32", c_add_u32_e32, &hz);
  const double co32 = run6("v_add_co_u32_e32 (VCC out)", c_add_co_e32, &hz);
  const double ci32 = run6("v_addc_co_u32_e32 (VCC in/out)", c_addc_co_e32, &hz);
  const double co64 = run6("v_add_co_u32_e64 (SGPR pair out)", c_add_co_e64, &hz);
  const double ci64 = run6("v_addc_co_u32_e64 (SGPR in/out)", c_addc_co_e64, &hz);
  run6("v_sub_co_u32_e32 (VCC out)", c_sub_co_e32, &hz);
  run6("v_subb_co_u32_e64 (SGPR in/out)", c_subb_co_e64, &hz);
  run6("add_co_e32 + addc_e32 pairs", c_pair_e32, &hz);
  const double la64 = run6("v_lshl_add_u64", c_lshl_add_u64, &hz);
  const double m1v = run6("v_mad_u64_u32 x, 1, acc (VCC)", c_mad64_x1_vcc, &hz);
  const double m1s = run6("v_mad_u64_u32 x, 1, acc (SGPR)", c_mad64_x1_sgpr, &hz);
  const double mf = run6("v_mad_u64_u32 x, y, acc", c_mad64_full, &hz);
  const double mi = run6("v_mad_u64_u32 x, inline const, acc", c_mad64_inline, &hz);
  const double msg = run6("v_mad_u64_u32 x, SGPR, acc", c_mad64_sgpr_op, &hz);
  run6("v_cndmask_b32_e64 (SGPR mask)", c_cndmask_e64, &hz);
  run6("v_cmp_lt_u32_e64 (SGPR out)", c_cmp_e64, &hz);
  // the shipped leaf sponge's static mix (3,791 v_mad_u64_u32, 1,161 VOP2 carry ops, 1,546 VOP3 carry ops; the rest
  // of its 6,512 VALU instructions are plain 32-bit ops): cycle-weighted mean against the flat 4-cycle assumption
  const double n_mad = 3791, n_e32 = 1161, n_e64 = 1546, n_other = 6512 - 3791 - 1161 - 1546;
  (void)mf;
  const double mix = (n_mad * 0.5 * (mi + msg) + n_e32 * 0.5 * (co32 + ci32) + n_e64 * 0.5 * (co64 + ci64) + (n_other > 0 ? n_other : 0) * add32) / 6512;
  printf("cycle-weighted mix of k_hash_leaves_wide's static instruction counts: %.2f cycles per wave instruction (flat assumption: 4.00)\n", mix);
  (void)la64; (void)m1v; (void)m1s;
  printf("\n-- residency and workgroup shape (the same instruction streams) --\n");
  for (int wg : {64, 256})
    for (int ps : {2, 4, 6, 8}) {
      run6("v_mad_u64_u32 x, y, acc", c_mad64_full, &hz, wg, ps);
      run6("add_co_e32 + addc_e32 pairs", c_pair_e32, &hz, wg, ps);
      run6("v_addc_co_u32_e64 (SGPR in/out)", c_addc_co_e64, &hz, wg, ps);
    }
  return 0;
}
