// Issue cost of the cross-lane VALU operations of gfx950, one to three waves per SIMD:
// 8 independent chains per wave, cycles from the event time at a fixed grid.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ void k(float *o, int iters) {
  float v[8];
  for (int i = 0; i < 8; i++) v[i] = threadIdx.x + i;
  float w = 1.0f;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#define OP(i)                                                                                                   \
  if (MODE == 0) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i])); \
  if (MODE == 1) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));          \
  if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v[i]));         \
  if (MODE == 3) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));   \
  if (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w));                                          \
  if (MODE == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(w));                                  \
  if (MODE == 6) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[i]), "+v"(w));                                  \
  if (MODE == 7) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[i]));       \
  if (MODE == 8) asm volatile("v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[i]));         \
  if (MODE == 9) asm volatile("v_add_f32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "+v"(v[i]));
      REP8(OP)
    }
  }
  float s = w;
  for (int i = 0; i < 8; i++) s += v[i];
  o[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE> void run(const char *n, int grid) {
  float *d; hipMalloc(&d, 16384 * 64 * 4);
  const int iters = 4000;
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, 10); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int t = 0; t < 3; t++) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  // ops per SIMD = grid/1024 waves * iters*64 ops ; time -> ns per op per SIMD
  double ops_per_simd = (double)grid / 1024.0 * iters * 64.0;
  printf("%-28s grid %5d: %.2f ns per op and SIMD (%.1f clk at 2.4 GHz)\n", n, grid, best * 1e6 / ops_per_simd, best * 1e6 / ops_per_simd * 2.4);
  hipFree(d);
}
int main() {
  for (int grid : {1024, 3072}) {
    run<4>("v_add_f32", grid);
    run<0>("v_add_f32_dpp quad_perm", grid);
    run<1>("v_add_f32_dpp row_shr:1", grid);
    run<2>("v_add_f32_dpp row_mirror", grid);
    run<7>("v_add_f32_dpp row_bcast:15", grid);
    run<8>("v_add_f32_dpp wave_shr:1", grid);
    run<3>("v_mov_b32_dpp quad_perm", grid);
    run<5>("v_permlane32_swap_b32", grid);
    run<6>("v_permlane16_swap_b32", grid);
    run<9>("v_add_f32_sdwa", grid);
  }
  return 0;
}
