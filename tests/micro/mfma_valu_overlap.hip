// Do the matrix pipe and the VALU of one SIMD run concurrently for different waves?
// Workgroups of 64 threads; even blocks issue MFMAs (16x16x4 f32, accumulating), odd blocks issue fp32 FMAs.
// grid 2048 puts two waves on every SIMD.  Compare: all-MFMA, all-FMA, mixed.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(64) k(float *o, int iters, int mode) {  // mode 0: all mfma, 1: all fma, 2: even mfma / odd fma
  const bool do_mfma = mode == 0 || (mode == 2 && ((blockIdx.x >> 10) & 1) == 0);  // blocks 0..1023 first wave of every SIMD, 1024..2047 second
  float acc = 0.f;
  if (do_mfma) {
    v4f d0 = {0, 0, 0, 0}, d1 = {0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, d1, 0, 0, 0);
      }
    }
    acc = d0[0] + d1[1];
  } else {
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = fmaf(v[i], 1.0001f, 0.5f);   // 128 fma per iteration
    }
    for (int i = 0; i < 8; i++) acc += v[i];
  }
  o[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
  float *d; hipMalloc(&d, 4096 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int grid : {1024, 2048}) for (int mode = 0; mode < 3; mode++) {
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, 10, mode); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("grid %d mode %s: %.3f ms (per iteration: 16 mfma or 128 fma per wave: %.1f ns)\n", grid,
           mode == 0 ? "all-mfma" : mode == 1 ? "all-fma " : "mixed   ", ms, ms * 1e6 / iters);
  }
}
