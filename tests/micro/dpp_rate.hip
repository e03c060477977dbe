// Issue cost of DPP and plain VALU ops for one wave per SIMD on gfx950 (cycles from s_memtime).
// CAUTION: dppadd() below is `v + update_dpp(0, v, ...)` with bound_ctrl off, which compiles to
// v_mov (old = 0) + v_mov_dpp + v_add -- three instructions, ~11.6 cycles -- not to the fused
// v_add_f32_dpp the kernels contain (4.4 cycles; see dpp_kinds.hip, which times that one).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dppadd(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// MODE 0: 8 independent chains of dpp adds; 1: 8 independent chains of plain adds; 2: one dependent dpp chain
// 3: one dependent 4-dpp chain + 12 independent fmas per "step"; 4: 12 independent fmas only; 5: 6 pk_fma indep
template <int MODE>
__global__ void k(float *o, long long *cyc, int iters) {
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
  float a = threadIdx.x * 0.5f, b = 1.0001f;
  float2 p[6];
  for (int i = 0; i < 6; i++) p[i] = make_float2(threadIdx.x + i, i);
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = dppadd<0xB1>(v[i]);
      } else if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = v[i] + b;
      } else if (MODE == 2) {
        v[0] = dppadd<0xB1>(v[0]); v[0] = dppadd<0x4E>(v[0]); v[0] = dppadd<0x141>(v[0]); v[0] = dppadd<0x140>(v[0]);
        v[0] = dppadd<0xB1>(v[0]); v[0] = dppadd<0x4E>(v[0]); v[0] = dppadd<0x141>(v[0]); v[0] = dppadd<0x140>(v[0]);
      } else if (MODE == 3) {
        float s = v[0];
        s = dppadd<0xB1>(s); s = dppadd<0x4E>(s); s = dppadd<0x141>(s); s = dppadd<0x140>(s);
#pragma unroll
        for (int i = 1; i < 8; i++) v[i] = fmaf(v[i], b, a);
#pragma unroll
        for (int i = 1; i < 6; i++) p[i].x = fmaf(p[i].x, b, a);
        v[0] = s * 0.5f;
      } else if (MODE == 4) {
#pragma unroll
        for (int i = 1; i < 8; i++) v[i] = fmaf(v[i], b, a);
#pragma unroll
        for (int i = 1; i < 6; i++) p[i].x = fmaf(p[i].x, b, a);
      } else if (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 6; i++) {
          float2 q = p[i];
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q) : "v"(make_float2(b, b)), "v"(make_float2(a, a)));
          p[i] = q;
        }
      }
    }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; i++) s += v[i];
  for (int i = 0; i < 6; i++) s += p[i].x + p[i].y;
  o[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *n, int nops, int grid) {
  float *d; long long *c; hipMalloc(&d, 8192 * 64 * 4); hipMalloc(&c, 8192 * 8);
  int iters = 2000;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, c, 10); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double per = (double)h[0] / (iters * 8.0);
  printf("%-44s grid %5d: %.2f clk per group of %d ops = %.2f clk/op ; wall %.1f ns/group\n", n, grid, per, nops, per / nops, ms * 1e6 / (iters * 8.0));
  hipFree(d); hipFree(c);
}
int main() {
  for (int grid : {1024, 2048, 4096}) {
    run<0>("8 independent dpp-add chains", 8, grid);
    run<1>("8 independent plain-add chains", 8, grid);
    run<2>("dependent dpp adds x8", 8, grid);
    run<3>("4 dep dpp + 12 indep fma + mul", 17, grid);
    run<4>("12 indep fma", 12, grid);
    run<5>("6 indep pk_fma", 6, grid);
  }
}
