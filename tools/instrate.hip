// Measures issue rates of individual gfx950 VALU instructions (inline asm, independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
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
  return 0;
}
