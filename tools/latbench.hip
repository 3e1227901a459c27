// Single-wave latency / issue cost of the instruction patterns the cooperative (latency-bound) permutations are made
// of: one wave on an otherwise idle chip, cycles from s_memtime.  `dep` = each instruction consumes the previous
// result, `ind4` = four independent chains interleaved.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint64_t u64;
typedef uint32_t u32;
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define BENCH(NAME, N_PER_ITER, SETUP, BODY)                                               \
  __global__ __launch_bounds__(64) void NAME(u64* out, int iters) {                        \
    u32 a = threadIdx.x * 2654435761u + 1, b = a ^ 0x9e3779b9u, c = a + 77, d = b * 3 + 1; \
    u64 x = a, y = b, z = c, w = d;                                                        \
    SETUP;                                                                                 \
    u64 t0 = __builtin_readcyclecounter();                                                 \
    for (int i = 0; i < iters; i++) { BODY; }                                              \
    u64 t1 = __builtin_readcyclecounter();                                                 \
    out[threadIdx.x] = a + b + c + d + x + y + z + w;                                      \
    if (threadIdx.x == 0) out[64] = t1 - t0;                                               \
  }
BENCH(k_mad_dep, 16, , REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc");))
BENCH(k_mad_ind4, 16, , REP4(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b) : "vcc");))
BENCH(k_mad_sgpr_dep, 16, u32 sa = __builtin_amdgcn_readfirstlane(a), REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "s"(sa), "v"(b) : "vcc");))
BENCH(k_add_dep, 16, , REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));))
BENCH(k_add_ind4, 16, , REP4(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x));))
BENCH(k_addco_dep, 16, , REP4(REP4(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "vcc");)))
BENCH(k_addco_nop_dep, 16, , REP4(REP4(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n s_nop 1\n v_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(a), "+v"(b) : "v"(c), "v"(d) : "vcc");)))
BENCH(k_lshladd64_dep, 16, , REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x) : "v"(y));))
BENCH(k_mullo_dep, 16, , REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));))
BENCH(k_mulhi_dep, 16, , REP16(asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(b));))
BENCH(k_bperm_dep, 16, u32 idx = ((threadIdx.x + 1) & 63) * 4, REP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(idx));))
BENCH(k_bperm_ind4, 16, u32 idx = ((threadIdx.x + 1) & 63) * 4, REP4(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(idx));))
BENCH(k_dpp_mov_dep, 16, , REP16(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a));))
BENCH(k_dpp_add_dep, 16, , REP16(asm volatile("s_nop 1\n v_add_u32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a));))
BENCH(k_dpp_add_nonop_ind, 16, , REP4(asm volatile("v_add_u32_dpp %0, %4, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %4, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %3 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(threadIdx.x));))
BENCH(k_readlane_mad, 16, , REP16(asm volatile("v_readlane_b32 s20, %1, 3\n v_mad_u64_u32 %0, vcc, s20, %2, %0" : "+v"(x) : "v"(a), "v"(b) : "vcc", "s20");))
BENCH(k_readlane4_mad4, 16, , REP4(asm volatile("v_readlane_b32 s20, %4, 1\n v_readlane_b32 s21, %4, 2\n v_readlane_b32 s22, %4, 3\n v_readlane_b32 s23, %4, 4\n v_mad_u64_u32 %0, vcc, s20, %5, %0\n v_mad_u64_u32 %1, vcc, s21, %5, %1\n v_mad_u64_u32 %2, vcc, s22, %5, %2\n v_mad_u64_u32 %3, vcc, s23, %5, %3" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b) : "vcc", "s20", "s21", "s22", "s23");))
BENCH(k_snop0, 16, , REP16(asm volatile("s_nop 0");))
BENCH(k_snop1, 16, , REP16(asm volatile("s_nop 1");))
BENCH(k_cndmask_dep, 16, , REP4(REP4(asm volatile("v_cmp_lt_u32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");)))
BENCH(k_ldsread_dep, 16, __shared__ u32 sh[64]; sh[threadIdx.x] = (threadIdx.x * 4 + 4) & 255; __syncthreads(); u32 p = threadIdx.x * 4, REP16(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(p));) a += p;)
BENCH(k_swizzle_dep, 16, , REP16(asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n s_waitcnt lgkmcnt(0)" : "+v"(a));))
BENCH(k_perml_dep, 16, , REP16(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));))

#include "../plonky2.5_amd/csrc/gl.h"
__global__ __launch_bounds__(64) void k_mulnc_dep(u64* out, int iters) {
  u64 x = threadIdx.x * 0x9E3779B97F4A7C15ull + 12345, y = x ^ 0xABCDEF0123456789ull;
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = gl::mul_nc(x, y);
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) out[64] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_mulnc_ind2(u64* out, int iters) {
  u64 x = threadIdx.x * 0x9E3779B97F4A7C15ull + 12345, y = x ^ 0xABCDEF0123456789ull, z = x + 99;
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) { x = gl::mul_nc(x, y); z = gl::mul_nc(z, y); }
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x + z;
  if (threadIdx.x == 0) out[64] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_mulc_dep(u64* out, int iters) {  // compiler form (no inline asm): reduce128
  u64 x = threadIdx.x * 0x9E3779B97F4A7C15ull + 12345, y = x ^ 0xABCDEF0123456789ull;
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = gl::reduce128(x * y, __umul64hi(x, y));
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) out[64] = t1 - t0;
}
__global__ __launch_bounds__(64) void k_addmod_dep(u64* out, int iters) {
  u64 x = threadIdx.x * 0x9E3779B97F4A7C15ull % gl::P, y = (x ^ 0xABCDEF0123456789ull) % gl::P;
  u64 t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 16; k++) x = gl::add(x, y);
  }
  u64 t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x;
  if (threadIdx.x == 0) out[64] = t1 - t0;
}

template <class K> void run(const char* name, K k, int per_iter) {
  u64* d; (void)hipMalloc(&d, 65 * 8);
  const int iters = 2000;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 10);
  (void)hipDeviceSynchronize();
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, iters);
  (void)hipDeviceSynchronize();
  u64 h[65]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-22s %8.2f cycles per instruction (or per unit)\n", name, (double)h[64] / ((double)iters * per_iter));
  (void)hipFree(d);
}
int main() {
  run("mad_u64_u32 dep", k_mad_dep, 16); run("mad_u64_u32 ind4", k_mad_ind4, 16); run("mad sgpr-src dep", k_mad_sgpr_dep, 16);
  run("v_add_u32 dep", k_add_dep, 16); run("v_add_u32 ind4", k_add_ind4, 16);
  run("add_co+addc pair dep", k_addco_dep, 16); run("add_co+nop+addc pair", k_addco_nop_dep, 16);
  run("lshl_add_u64 dep", k_lshladd64_dep, 16); run("mul_lo dep", k_mullo_dep, 16); run("mul_hi dep", k_mulhi_dep, 16);
  run("bpermute dep", k_bperm_dep, 16); run("bpermute ind4 (per 1)", k_bperm_ind4, 16);
  run("dpp mov dep (+nop1)", k_dpp_mov_dep, 16); run("dpp add dep (+nop1)", k_dpp_add_dep, 16); run("dpp add ind4", k_dpp_add_nonop_ind, 16);
  run("readlane+mad", k_readlane_mad, 16); run("4 readlane+4 mad (per pair)", k_readlane4_mad4, 16);
  run("s_nop 0", k_snop0, 16); run("s_nop 1", k_snop1, 16); run("cmp+cndmask pair dep", k_cndmask_dep, 16);
  run("ds_read_b32 dep", k_ldsread_dep, 16); run("ds_swizzle dep", k_swizzle_dep, 16); run("permlane32_swap dep", k_perml_dep, 16);
  run("gl::mul_nc dep", k_mulnc_dep, 16); run("gl::mul_nc ind2", k_mulnc_ind2, 16); run("compiler mul dep", k_mulc_dep, 16);
  run("gl::add dep", k_addmod_dep, 16);
  return 0;
}
