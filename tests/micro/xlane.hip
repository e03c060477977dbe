// Semantics + latency of v_permlane16_swap and DPP wave_shr:1 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sem(float *o) {
  int l = threadIdx.x;
  float v = (float)l;
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[l] = __builtin_bit_cast(float, r[0]);
  o[64 + l] = __builtin_bit_cast(float, r[1]);
  float w = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -1.0f), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
  o[128 + l] = w;
}
template <int MODE>
__global__ void lat(float *o, int iters) {
  float v = threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int r = 0; r < 32; r++) {
      if (MODE == 0) {
        unsigned a = __builtin_bit_cast(unsigned, v), b = a;
        auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        v = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
      } else if (MODE == 1) {
        v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false)) + 1.0f;
      } else {
        v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false)) + 1.0f;
      }
    }
  }
  o[blockIdx.x * 64 + threadIdx.x] = v;
}
template <int MODE> void run(const char *n) {
  float *d; hipMalloc(&d, 1024 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int iters = 20000;
  hipLaunchKernelGGL(lat<MODE>, dim3(1024), dim3(64), 0, 0, d, 10); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(lat<MODE>, dim3(1024), dim3(64), 0, 0, d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%s: %.1f ns per dependent (op + add) = %.1f cycles @2.1GHz\n", n, ms * 1e6 / (iters * 32.0), ms * 1e6 / (iters * 32.0) * 2.1);
}
int main() {
  float *d; hipMalloc(&d, 192 * 4); float h[192];
  hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("swap r0: "); for (int i = 0; i < 64; i += 4) printf("%g ", h[i]); printf("\nswap r1: "); for (int i = 0; i < 64; i += 4) printf("%g ", h[64 + i]);
  printf("\nwave_shr: "); for (int i = 0; i < 64; i += 1) if (i < 3 || (i > 14 && i < 19) || (i > 30 && i < 35) || i > 61) printf("[%d]=%g ", i, h[128 + i]); printf("\n");
  run<0>("permlane16_swap + add"); run<1>("dpp wave_shr:1 + add"); run<2>("dpp row_shr:1 + add");
}
