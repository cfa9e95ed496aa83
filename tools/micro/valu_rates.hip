// Micro-benchmark (not part of the product): issue cost of the VALU instructions the rates kernel is made of, in
// cycles per wave64 instruction with the SIMDs saturated (8 waves per SIMD, long independent chains).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3;
  for (int it = 0; it < iters; it++) {
    if (OP == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 1) { REP16(asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 2) { REP16(asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 3) { REP16(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
    if (OP == 4) { REP16(asm volatile("v_cvt_i32_f64 %0, %4\n v_cvt_i32_f64 %1, %5\n v_cvt_i32_f64 %2, %6\n v_cvt_i32_f64 %3, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if (OP == 5) { REP16(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 6) { REP16(asm volatile("v_fract_f64 %0, %0\n v_fract_f64 %1, %1\n v_fract_f64 %2, %2\n v_fract_f64 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 7) { REP16(asm volatile("v_max_f64 %0, %0, %1\n v_max_f64 %1, %1, %2\n v_max_f64 %2, %2, %3\n v_max_f64 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 8) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));) }
    if (OP == 9) { REP16(asm volatile("v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %2, %3\n v_cmp_gt_f64 vcc, %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
    if (OP == 10) { REP16(asm volatile("v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 11) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 3, %1\n v_lshl_add_u64 %1, %1, 3, %2\n v_lshl_add_u64 %2, %2, 3, %3\n v_lshl_add_u64 %3, %3, 3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 12) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");) }
    if (OP == 13) { REP16(asm volatile("v_frexp_exp_i32_f64 %0, %4\n v_frexp_exp_i32_f64 %1, %5\n v_frexp_exp_i32_f64 %2, %6\n v_frexp_exp_i32_f64 %3, %7" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
    if (OP == 14) { REP16(asm volatile("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %5\n v_ldexp_f64 %2, %2, %6\n v_ldexp_f64 %3, %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;
}

template <int OP>
void run(const char *name) {
  double *d;
  const int blocks = 256 * 8, iters = 2000; // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipMalloc(&d, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: 8 waves x iters x 64; cycles at 2.4 GHz nominal
  const double instr_per_simd = 8.0 * iters * 64.0;
  printf("%-22s %7.3f ms  %6.2f cycles per wave-instruction at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
  hipFree(d);
}

int main() {
  run<0>("v_fma_f64"); run<1>("v_mul_f64"); run<2>("v_add_f64"); run<3>("v_cvt_f64_i32"); run<4>("v_cvt_i32_f64");
  run<5>("v_rcp_f64"); run<6>("v_fract_f64"); run<7>("v_max_f64"); run<8>("v_add_u32"); run<9>("v_cmp_gt_f64");
  run<10>("v_mov_b64"); run<11>("v_lshl_add_u64"); run<12>("v_cndmask_b32"); run<13>("v_frexp_exp_i32_f64"); run<14>("v_ldexp_f64");
  return 0;
}
