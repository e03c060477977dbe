// semantics + cost of v_mfma_f32_16x16x4_f32 used as a cross-lane column sum / prefix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void sem(float *o) {
  int l = threadIdx.x;
  float v = (float)(l + 1);
  v4f c = {0.f, 0.f, 0.f, 0.f};
  v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, v, c, 0, 0, 0);
  o[l] = d[0]; o[64 + l] = d[3];
  float a = ((l / 16) <= (l % 16) / 4) ? 1.0f : 0.0f;
  v4f e = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v, c, 0, 0, 0);
  o[128 + l] = e[0]; o[192 + l] = e[2];
}
template <int MODE> __global__ void lat(float *o, long long *cyc, int iters) {
  float v = threadIdx.x, w = 1.0f, u[8];
  for (int i = 0; i < 8; i++) u[i] = threadIdx.x + i;
  v4f c = {0.f, 0.f, 0.f, 0.f};
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (MODE == 0) { v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(w, v, c, 0, 0, 0); v = d[0] * 0.25f; }      // dependent chain
      else if (MODE == 1) { v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(w, u[r], c, 0, 0, 0); u[r] = d[0] * 0.25f; } // 8 independent
      else { v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(w, u[r], c, 0, 0, 0); u[r] = d[0] * 0.25f;
#pragma unroll
        for (int i = 0; i < 8; i++) u[(r + 1 + i) & 7] = fmaf(u[(r + 1 + i) & 7], 1.0001f, 0.5f); }  // 1 mfma + 8 fma
    }
  }
  long long t1 = clock64();
  float s = v; for (int i = 0; i < 8; i++) s += u[i];
  o[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *n, int grid) {
  float *d; long long *c; hipMalloc(&d, 8192 * 64 * 4); hipMalloc(&c, 8192 * 8);
  int iters = 2000;
  hipLaunchKernelGGL(lat<MODE>, dim3(grid), dim3(64), 0, 0, d, c, 10); hipDeviceSynchronize();
  hipLaunchKernelGGL(lat<MODE>, dim3(grid), dim3(64), 0, 0, d, c, iters); hipDeviceSynchronize();
  long long h[4]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-40s grid %d: %.1f clk per op-group\n", n, grid, (double)h[0] / (iters * 8.0));
}
int main() {
  float *d; hipMalloc(&d, 256 * 4); float h[256];
  hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("colsum d[0]: "); for (int i = 0; i < 64; i += 5) printf("[%d]=%g ", i, h[i]); printf("\n  expect lane l: sum_k (16k + l%%16 + 1) = 4*(l%%16+1) + 96\n");
  printf("colsum d[3]: "); for (int i = 0; i < 64; i += 5) printf("[%d]=%g ", i, h[64 + i]); printf("\n");
  printf("prefix e[0]: "); for (int i = 0; i < 64; i += 3) printf("[%d]=%g ", i, h[128 + i]); printf("\n  expect lane (k,j): sum_{k'<=k} (16k' + j + 1)\n");
  printf("prefix e[2]: "); for (int i = 0; i < 64; i += 3) printf("[%d]=%g ", i, h[192 + i]); printf("\n");
  for (int g : {1024, 2048}) { run<0>("dependent mfma + mul", g); run<1>("8 independent mfma + mul", g); run<2>("mfma + 8 fma", g); }
}
