// tests/micro/xchg.hip -- the 8 x 8 float2 transpose between a lane's low three bits and its register index
// (the second exchange of the 512-point radix-8 plan) two ways, at the front kernel's occupancy (two waves per
// SIMD): through LDS (8 ds_write_b64 + 8 ds_read_b64, the product's form) and in registers (v_cndmask_b32_dpp
// for lane bits 0 and 1, masked v_mov_b32_dpp for bit 2: 56 VALU instructions).  Each iteration also runs one
// radix-8 butterfly (26 packed instructions) so that the exchange competes with real VALU work.
//   hipcc -O3 --offload-arch=gfx950 -I radiodsp_sdr_rx_amd/csrc tests/micro/xchg.hip -o /tmp/xchg && /tmp/xchg
#include <hip/hip_runtime.h>
#pragma clang diagnostic ignored "-Wunused-value"
#include <cstdio>
#include <vector>
#include "rdsp_fft.h"
using namespace rdsp;

template <int CTRL>
__device__ __forceinline__ float dppmov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// lane bit B (0 or 1) <-> register bit B, one 32-bit plane of the eight registers
template <int B>
__device__ __forceinline__ void xstage_q(float (&r)[8], bool bset) {
  constexpr int CTRL = B == 0 ? 0xB1 : 0x4E;  // quad_perm [1,0,3,2] / [2,3,0,1]
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (k & (1 << B)) continue;
    const float lo = r[k], hi = r[k | (1 << B)];
    const float plo = dppmov<CTRL>(lo), phi = dppmov<CTRL>(hi);
    r[k] = bset ? phi : lo;            // the compiler folds the DPP move into v_cndmask_b32_dpp
    r[k | (1 << B)] = bset ? hi : plo;
  }
}
// lane bit 2 <-> register bit 2: row_shr:4 / row_shl:4 with bank masks
__device__ __forceinline__ void xstage_b2(float (&r)[8]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int lo = __builtin_bit_cast(int, r[k]), hi = __builtin_bit_cast(int, r[k + 4]);
    // lanes with bit 2 set (banks 1, 3) take the partner's hi into lo; lanes without take the partner's lo into hi
    const int nlo = __builtin_amdgcn_update_dpp(lo, hi, 0x114, 0xF, 0xA, false);  // row_shr:4, banks 1 and 3
    const int nhi = __builtin_amdgcn_update_dpp(hi, lo, 0x104, 0xF, 0x5, false);  // row_shl:4, banks 0 and 2
    r[k] = __builtin_bit_cast(float, nlo);
    r[k + 4] = __builtin_bit_cast(float, nhi);
  }
}

// the same by hand: v_cndmask_b32_dpp (VOP2: DPP on src0, the mask in VCC), 16 per stage and plane pair;
// four at a time behind one s_mov_b64 vcc (the constraint syntax has no way to name VCC as an input)
#define XQ4(CTRLSTR, o0, o1, o2, o3, a0, a1, a2, a3, b0, b1, b2, b3, m)                                  \
  asm("s_mov_b64 vcc, %12\n\t"                                                                           \
      "v_cndmask_b32_dpp %0, %4, %8, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                       \
      "v_cndmask_b32_dpp %1, %5, %9, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                       \
      "v_cndmask_b32_dpp %2, %6, %10, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf\n\t"                      \
      "v_cndmask_b32_dpp %3, %7, %11, vcc " CTRLSTR " row_mask:0xf bank_mask:0xf"                          \
      : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)                                                       \
      : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "s"(m) : "vcc")
template <int B>
__device__ __forceinline__ void xstage_q_asm(float (&r)[8], unsigned long long mset, unsigned long long mclear) {
  // pairs (lo, hi): B = 0: (0,1) (2,3) (4,5) (6,7); B = 1: (0,2) (1,3) (4,6) (5,7)
  constexpr int L0 = 0, L1 = B == 0 ? 2 : 1, L2 = 4, L3 = B == 0 ? 6 : 5, D = 1 << B;
  float n0, n1, n2, n3, h0, h1, h2, h3;
  if constexpr (B == 0) {
    XQ4("quad_perm:[1,0,3,2]", n0, n1, n2, n3, r[L0 + D], r[L1 + D], r[L2 + D], r[L3 + D], r[L0], r[L1], r[L2], r[L3], mclear);
    XQ4("quad_perm:[1,0,3,2]", h0, h1, h2, h3, r[L0], r[L1], r[L2], r[L3], r[L0 + D], r[L1 + D], r[L2 + D], r[L3 + D], mset);
  } else {
    XQ4("quad_perm:[2,3,0,1]", n0, n1, n2, n3, r[L0 + D], r[L1 + D], r[L2 + D], r[L3 + D], r[L0], r[L1], r[L2], r[L3], mclear);
    XQ4("quad_perm:[2,3,0,1]", h0, h1, h2, h3, r[L0], r[L1], r[L2], r[L3], r[L0 + D], r[L1 + D], r[L2 + D], r[L3 + D], mset);
  }
  r[L0] = n0; r[L1] = n1; r[L2] = n2; r[L3] = n3;
  r[L0 + D] = h0; r[L1 + D] = h1; r[L2 + D] = h2; r[L3 + D] = h3;
}
__device__ __forceinline__ void xstage_b2_asm(float (&r)[8]) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float t = r[k + 4];  // (lanes without bit 2 still need their old hi as the partner's source)
    float lo = r[k], hi = r[k + 4];
    asm("v_mov_b32_dpp %0, %1 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(hi) : "v"(lo));
    asm("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(lo) : "v"(t));
    r[k] = lo; r[k + 4] = hi;
  }
}

template <int MODE>
__global__ void __launch_bounds__(64, 2) xk(float2 *buf, int iters) {
  __shared__ float2 w[576];  // one wave per workgroup, the padded map of the 512-point plan's second exchange
  const int lane = threadIdx.x;
  float2 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = buf[(blockIdx.x * 64 + lane) * 8 + k];
  const int a = lane >> 3, b = lane & 7;
  for (int it = 0; it < iters; it++) {
    Dft<8, false>::run(v);
    if constexpr (MODE == 0) {
      // lane (a, b) element k at 64 a + b + 8 k (map i + (i >> 3)); then lane reads 8 lane + e
#pragma unroll
      for (int k = 0; k < 8; k++) { const int i = 64 * a + b + 8 * k; w[i + (i >> 3)] = v[k]; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int e = 0; e < 8; e++) { const int i = 8 * lane + e; v[e] = lds_ld(&w[i + (i >> 3)]); }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else if constexpr (MODE == 1) {
      float re[8], im[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { re[k] = v[k].x; im[k] = v[k].y; }
      xstage_q<0>(re, lane & 1); xstage_q<0>(im, lane & 1);
      xstage_q<1>(re, lane & 2); xstage_q<1>(im, lane & 2);
      xstage_b2(re); xstage_b2(im);
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = make_float2(re[k], im[k]);
    } else {
      float re[8], im[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { re[k] = v[k].x; im[k] = v[k].y; }
      asm volatile("s_nop 1");
      xstage_q_asm<0>(re, 0xAAAAAAAAAAAAAAAAull, 0x5555555555555555ull); xstage_q_asm<0>(im, 0xAAAAAAAAAAAAAAAAull, 0x5555555555555555ull);
      xstage_q_asm<1>(re, 0xCCCCCCCCCCCCCCCCull, 0x3333333333333333ull); xstage_q_asm<1>(im, 0xCCCCCCCCCCCCCCCCull, 0x3333333333333333ull);
      xstage_b2_asm(re); xstage_b2_asm(im);
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = make_float2(re[k], im[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; k++) buf[(blockIdx.x * 64 + lane) * 8 + k] = v[k];
}

// the two forms move the same data: check on a labelled pattern, one iteration without the butterfly effect
template <int MODE>
__global__ void xcheck(float2 *buf) {
  __shared__ float2 w[576];
  const int lane = threadIdx.x, a = lane >> 3, b = lane & 7;
  float2 v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = make_float2((float)(lane * 8 + k), -(float)(lane * 8 + k));
  if constexpr (MODE == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) { const int i = 64 * a + b + 8 * k; w[i + (i >> 3)] = v[k]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) { const int i = 8 * lane + e; v[e] = w[i + (i >> 3)]; }
  } else if constexpr (MODE == 1) {
    float re[8], im[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { re[k] = v[k].x; im[k] = v[k].y; }
    xstage_q<0>(re, lane & 1); xstage_q<0>(im, lane & 1);
    xstage_q<1>(re, lane & 2); xstage_q<1>(im, lane & 2);
    xstage_b2(re); xstage_b2(im);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = make_float2(re[k], im[k]);
  } else {
    float re[8], im[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { re[k] = v[k].x; im[k] = v[k].y; }
    asm volatile("s_nop 1");
    xstage_q_asm<0>(re, 0xAAAAAAAAAAAAAAAAull, 0x5555555555555555ull); xstage_q_asm<0>(im, 0xAAAAAAAAAAAAAAAAull, 0x5555555555555555ull);
    xstage_q_asm<1>(re, 0xCCCCCCCCCCCCCCCCull, 0x3333333333333333ull); xstage_q_asm<1>(im, 0xCCCCCCCCCCCCCCCCull, 0x3333333333333333ull);
    xstage_b2_asm(re); xstage_b2_asm(im);
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = make_float2(re[k], im[k]);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) buf[lane * 8 + k] = v[k];
}

int main() {
  const int wgs = 256 * 8, iters = 20000;
  float2 *d;
  hipMalloc(&d, sizeof(float2) * wgs * 64 * 8);
  hipMemset(d, 0, sizeof(float2) * wgs * 64 * 8);
  std::vector<float2> a(512), b(512);
  std::vector<float2> c(512);
  xcheck<0><<<1, 64>>>(d); hipMemcpy(a.data(), d, sizeof(float2) * 512, hipMemcpyDeviceToHost);
  xcheck<1><<<1, 64>>>(d); hipMemcpy(b.data(), d, sizeof(float2) * 512, hipMemcpyDeviceToHost);
  xcheck<2><<<1, 64>>>(d); hipMemcpy(c.data(), d, sizeof(float2) * 512, hipMemcpyDeviceToHost);
  int bad = 0, bad2 = 0;
  for (int i = 0; i < 512; i++) { bad += (a[i].x != b[i].x || a[i].y != b[i].y); bad2 += (a[i].x != c[i].x || a[i].y != c[i].y); }
  // the LDS form is lane (a,b) element k -> lane a*8+k element b: the same transpose inside each group of 8 lanes
  printf("register transposes vs LDS exchange: %d (compiler) and %d (hand-written) of 512 elements differ\n", bad, bad2);
  hipMemset(d, 0, sizeof(float2) * wgs * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[3] = {"LDS (8 ds_write_b64 + 8 ds_read_b64)", "registers, compiler (121 VALU)    ", "registers, by hand (56 VALU)      "};
  for (int rep = 0; rep < 3; rep++) for (int mode = 0; mode < 3; mode++) {
    hipEventRecord(e0);
    if (mode == 0) xk<0><<<wgs, 64>>>(d, iters); else if (mode == 1) xk<1><<<wgs, 64>>>(d, iters); else xk<2><<<wgs, 64>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: %.3f ms, %.1f ns per butterfly + exchange per wave (8 waves per CU)\n", names[mode], ms, ms * 1e6 / iters);
  }
  return 0;
}
