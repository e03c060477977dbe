// Energy per operation by kind: each mode runs one instruction kind in long unrolled loops on every SIMD
// (4 waves per SIMD) for ~2.5 s while tests/micro/energy_ops.sh samples clock and package power beside it.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tests/micro/energy_ops.hip -o /tmp/energy_ops
//   /tmp/energy_ops <mode> [seconds]     modes: idle pkfma pkadd fma add dpp cvt ldsr ldsw
// Prints the wave-instructions per second it reached; energy per lane-operation = (W - W_idle-loop) / (rate * 64).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int UNROLL = 16, INNER = 4096;

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int reps) {
  __shared__ v2f lds[256 * 4];
  const int t = threadIdx.x;
  v2f a[8];
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; i++) { a[i] = v2f{1.0f + 1e-3f * i + 1e-6f * t, 0.5f}; s[i] = 1.0f + 1e-3f * i; }
  const v2f m = {0.999999f, 1.000001f}, c = {1e-7f, -1e-7f};
  lds[t] = a[0]; lds[t + 256] = a[1]; lds[t + 512] = a[2]; lds[t + 768] = a[3];
  __syncthreads();
  for (int r = 0; r < reps; r++) {
#pragma unroll 1
    for (int it = 0; it < INNER / UNROLL; it++) {
#pragma unroll
      for (int u = 0; u < UNROLL / 8; u++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          if constexpr (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
          if constexpr (MODE == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
          if constexpr (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(m[0]), "v"(c[0]));
          if constexpr (MODE == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(c[0]));
          if constexpr (MODE == 5) asm volatile("v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(s[i]) : "v"(s[(i + 4) & 7]));
          if constexpr (MODE == 6) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(s[i]) : "v"(t + i));
          if constexpr (MODE == 7) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(a[i]) : "v"(t * 8), "n"(2048 * (i & 3)));
          if constexpr (MODE == 8) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(t * 8), "v"(a[i]), "n"(2048 * (i & 3)) : "memory");
        }
        if constexpr (MODE == 7 || MODE == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if constexpr (MODE == 0) __builtin_amdgcn_s_sleep(8);
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) acc += a[i][0] + a[i][1] + s[i];
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
double run(float *d, double seconds, const char *name) {
  const int grid = 256 * 4; /* 4 workgroups of 4 waves per CU: 4 waves per SIMD */
  hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, 1);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  const int reps = 64;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, reps);
    hipDeviceSynchronize();
    launches++;
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const double winst = (double)launches * reps * INNER * (double)grid * 4.0; /* wave-instructions of the kind */
  printf("%s: %.2f s, %.3e wave-instructions/s of the kind (%.3e lane-ops/s)\n", name, dt, MODE ? winst / dt : 0.0, MODE ? winst / dt * 64 : 0.0);
  return winst / dt;
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "pkfma";
  const double sec = argc > 2 ? atof(argv[2]) : 2.5;
  float *d;
  hipMalloc(&d, 64);
  if (!strcmp(mode, "idle")) run<0>(d, sec, mode);
  else if (!strcmp(mode, "pkfma")) run<1>(d, sec, mode);
  else if (!strcmp(mode, "pkadd")) run<2>(d, sec, mode);
  else if (!strcmp(mode, "fma")) run<3>(d, sec, mode);
  else if (!strcmp(mode, "add")) run<4>(d, sec, mode);
  else if (!strcmp(mode, "dpp")) run<5>(d, sec, mode);
  else if (!strcmp(mode, "cvt")) run<6>(d, sec, mode);
  else if (!strcmp(mode, "ldsr")) run<7>(d, sec, mode);
  else if (!strcmp(mode, "ldsw")) run<8>(d, sec, mode);
  else { printf("unknown mode\n"); return 1; }
  return 0;
}
