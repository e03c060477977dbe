// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs DPP adds on gfx950,
// as a function of waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, int iters, float a, float b) {
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;
  float2 p[8];
  for (int i = 0; i < 8; i++) p[i] = make_float2(v[2 * i], v[2 * i + 1]);
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(make_float2(a, a)), "v"(make_float2(b, b)));
    } else if (MODE == 2) {  // pk_fma with scalar (SGPR) broadcast operand like the FIR
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++)
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(make_float2(a, a)), "v"(make_float2(b, b)));
    } else if (MODE == 3) {  // dependent chain of v_fma
#pragma unroll
      for (int r = 0; r < 64; r++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[0]) : "v"(a), "v"(b));
    } else if (MODE == 4) {  // dependent chain of DPP adds
#pragma unroll
      for (int r = 0; r < 64; r++) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1" : "+v"(v[0]));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; i++) s += v[i];
  for (int i = 0; i < 8; i++) s += p[i].x + p[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
void run(const char *name, int flops_per_iter_per_lane) {
  float *d;
  hipMalloc(&d, 256 * 4 * 16 * 64 * 4);
  for (int wps : {1, 2, 4, 8}) {
    int grid = 256 * 4 * wps, iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_wave = (double)iters * 64;  // 64 instructions per iteration in every mode
    double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
    double tflops = (double)grid * 64 * iters * flops_per_iter_per_lane / (ms * 1e-3) / 1e12;
    printf("%-28s waves/SIMD %d: %.3f ms  %.2f ns per wave-instr per SIMD (%.2f cycles @2.1GHz)  %.1f TFLOP/s\n", name, wps, ms,
           ns_per_instr_per_simd, ns_per_instr_per_simd * 2.1, tflops);
  }
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32 indep x16", 64 * 2);
  run<1>("v_pk_fma_f32 indep x8", 32 * 4 * 2);   // only 32 instr/iter: report per instr below
  run<2>("v_pk_fma_f32 op_sel bcast x8", 32 * 4 * 2);
  run<3>("v_fma_f32 dependent", 64 * 2);
  run<4>("v_add_f32_dpp dependent", 64);
  return 0;
}
