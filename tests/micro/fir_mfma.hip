// Prototype: the 256-tap /4 decimating FIR of one channel-chunk (1024 in -> 256 out, complex)
// as a GEMM on the matrix pipe:  y[16a+b] = sum_d x[64a+d] h[4b-d],  d in [-256, 63]
//   A[a][d] = x[64a + d]  (the input, reshaped; LDS row stride padded by 2 float2)
//   B[d][b] = h[4b - d]   (banded Toeplitz of the taps, read from a zero-padded tap line)
// 80 K-slices of v_mfma_f32_16x16x4_f32 per component.  Checks against a direct FIR and times it.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int PADQ = 2;                       // float2 of padding per 64 samples
__host__ __device__ inline int xaddr(int idx) { int u = idx + 256; return u + (u >> 6) * PADQ; }  // idx in [-256, 1023]
constexpr int XL_N = 1280 + (1280 / 64) * PADQ;
constexpr int HZ_LO = 64, HZ_N = 64 + 320;      // hz[t + HZ_LO], t in [-64, 319]

template <int MODE, bool CH4>  // 0: MFMA FIR, 1: only the LDS reads (no mfma), 2: mfma without LDS reads; CH4: four accumulator chains
__global__ void __launch_bounds__(64, 2) k(const float2 *x_all, const float *h, float2 *y_all, int n_chunks, long long *cyc, int reps) {
  __shared__ __attribute__((aligned(16))) float2 xl[XL_N];
  __shared__ float hz[HZ_N];
  const int l = threadIdx.x, i = l & 15, kq = l >> 4;
  const float2 *x = x_all + (size_t)blockIdx.x * (256 + 1024 * (size_t)n_chunks);
  float2 *y = y_all + (size_t)blockIdx.x * 256 * (size_t)n_chunks;
  for (int t = l; t < HZ_N; t += 64) { int tt = t - HZ_LO; hz[t] = (tt >= 0 && tt < 256) ? h[tt] : 0.f; }
  for (int t = l; t < 256; t += 64) xl[xaddr(t - 256)] = x[t];
  long long t0 = clock64();
  for (int c = 0; c < n_chunks; c++) {
    for (int t = l; t < 1024; t += 64) xl[xaddr(t)] = x[256 + 1024 * c + t];
    __syncthreads();
    v4f dre = {0, 0, 0, 0}, dim = {0, 0, 0, 0}, dre1 = {0, 0, 0, 0}, dim1 = {0, 0, 0, 0};
    for (int rep = 0; rep < reps; rep++) {   // reps > 1: time the FIR itself, the chunk load amortised
    if (rep) { dre *= 0.f; dim *= 0.f; dre1 *= 0.f; dim1 *= 0.f; }
    const float *bp = hz + HZ_LO + (4 * i - kq + 256);  // - 4s per slice
#pragma unroll 8
    for (int s = 0; s < 80; s++) {
      float2 a; float b;
      if (MODE != 2) { a = xl[xaddr(64 * i + kq - 256 + 4 * s)]; b = bp[-4 * s]; } else { a = make_float2(1.f + s, 2.f); b = 0.5f; }
      if (MODE != 1) {
        if (CH4 && (s & 1)) {
          dre1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, dre1, 0, 0, 0);
          dim1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, dim1, 0, 0, 0);
        } else {
          dre = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, dre, 0, 0, 0);
          dim = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, dim, 0, 0, 0);
        }
      } else { dre[0] += a.x * b; dim[0] += a.y * b; }
    }
    if (CH4) { dre += dre1; dim += dim1; }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) y[256 * c + 64 * kq + 16 * r + i] = make_float2(dre[r], dim[r]);
    __syncthreads();
    for (int t = l; t < 256; t += 64) xl[xaddr(t - 256)] = xl[xaddr(768 + t)];   // history
    __syncthreads();
  }
  if (l == 0) cyc[blockIdx.x] = clock64() - t0;
}

int main() {
  const int nch = 4096, nck = 16;
  std::vector<float2> hx((size_t)nch * (256 + 1024 * nck));
  std::vector<float> hh(256);
  srand(1);
  for (auto &v : hx) v = make_float2((rand() % 2001 - 1000) / 1000.f, (rand() % 2001 - 1000) / 1000.f);
  for (int t = 0; t < 256; t++) hh[t] = sinf(0.05f * (t - 127.5f)) / (0.05f * (t - 127.5f) + 1e-9f) * 0.02f;
  float2 *dx, *dy; float *dh; long long *dc;
  hipMalloc(&dx, hx.size() * 8); hipMalloc(&dy, (size_t)nch * 256 * nck * 8); hipMalloc(&dh, 1024); hipMalloc(&dc, nch * 8);
  hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dh, hh.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((k<0, false>), dim3(nch), dim3(64), 0, 0, dx, dh, dy, nck, dc, 1); hipDeviceSynchronize();
  std::vector<float2> hy((size_t)256 * nck);
  hipMemcpy(hy.data(), dy + (size_t)7 * 256 * nck, hy.size() * 8, hipMemcpyDeviceToHost);
  const float2 *x7 = hx.data() + (size_t)7 * (256 + 1024 * nck) + 256;   // sample 0 of the stream
  double worst = 0, ymax = 0;
  for (int m = 0; m < 256 * nck; m += 7) {
    double ar = 0, ai = 0;
    for (int t = 0; t < 256; t++) { ar += (double)hh[t] * x7[4 * m - t].x; ai += (double)hh[t] * x7[4 * m - t].y; }
    worst = fmax(worst, fmax(fabs(ar - hy[m].x), fabs(ai - hy[m].y))); ymax = fmax(ymax, fabs(ar));
  }
  printf("max abs err %.3e (max |y| %.3f)\n", worst, ymax);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char *name) {
    float ms1 = 0, ms9 = 0;
    for (int reps : {1, 9}) {
      hipLaunchKernelGGL(kern, dim3(nch), dim3(64), 0, 0, dx, dh, dy, nck, dc, reps); hipDeviceSynchronize();
      hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(nch), dim3(64), 0, 0, dx, dh, dy, nck, dc, reps); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); (reps == 1 ? ms1 : ms9) = ms;
    }
    const double per = (ms9 - ms1) / 8.0;   // one FIR pass over all chunks
    printf("%-28s %.3f ms per pass (%d ch x %d chunks) -> %.0f cycles per chunk per SIMD @2.25 GHz, FIR-only rate %.1f Gsamples/s\n",
           name, per, nch, nck, per * 1e-3 * 2.25e9 / ((double)nch * nck / 1024.0), (double)nch * nck * 1024 / per / 1e6);
  };
  run(k<0, false>, "mfma FIR"); run(k<0, true>, "mfma FIR, 4 chains"); run(k<1, false>, "LDS reads only"); run(k<2, false>, "mfma only"); run(k<2, true>, "mfma only, 4 chains");
  return worst < 1e-4 ? 0 : 1;
}
