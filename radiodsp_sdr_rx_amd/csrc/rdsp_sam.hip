/*
 * rdsp_sam.hip -- SAMmode (CTL:384-391): the AudioSDR engine's synchronous AM detector, a
 * second-order PLL on the filtered base band (arithmetic build-defined,
 * DESIGN.md 6e).  Runs between the front kernel and the tail stage for the channels of
 * SAM groups.
 */
#include "rdsp_wave.h"

using namespace rdsp;

namespace {

/* ---- SAM demodulator: second-order PLL on the filtered base band --------------------
 * SAMmode (CTL:384-391) is an AudioSDR demodulator, build-defined here:
 *   corr0 = Re(y e^{-j phs}), corr1 = Im(y e^{-j phs}), det = atan2(corr1, corr0) weighted
 *   by |y|^2/(|y|^2 + 1e-6), omega += g2 det (clamped), phs += previous (g1 det + omega),
 *   audio = corr0 - dc.
 * The recursion is serial per sample, so a channel is one lane (64 channels per wave)
 * and the kernel is bound by the length of the dependent chain.  To keep that chain
 * short the oscillator is carried as a unit phasor rotated by the loop filter output
 * (known one sample ahead, so its sine/cosine are off the chain) instead of
 * sincos(phs), and the detector uses a short arctangent (two rcp, four fma; 2e-7):
 * ~25 dependent operations per sample against ~80 with the library calls (measured
 * 4.0 -> see DESIGN.md 6e).  phs is rebuilt from the phasor when the state is saved. */
__device__ __forceinline__ float sam_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx); /* [0, 1] */
  /* atan on [0, 1]: above tan(pi/8) fold with atan(t) = pi/4 + atan((t - 1)/(t + 1)) */
  const bool hi = t > 0.41421356237f;
  const float u = hi ? (t - 1.0f) * __builtin_amdgcn_rcpf(t + 1.0f) : t;
  const float z = u * u;
  float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  float r = fmaf(p * z, u, u) + (hi ? 0.78539816339744831f : 0.0f);
  r = (ay > ax) ? 1.57079632679489662f - r : r;
  r = (x < 0.0f) ? 3.14159265358979324f - r : r;
  return (y < 0.0f) ? -r : r;
}
/* sine and cosine of a small angle (|d| < 0.6 rad: the loop filter output is clamped
 * to +-2 kHz at 24 kHz plus g1*pi), Taylor to d^9 / d^8: 1e-9 */
__device__ __forceinline__ void sam_sincos_small(float d, float *sn, float *cs) {
  const float d2 = d * d;
  float s = fmaf(d2, 2.7557319e-6f, -1.9841270e-4f);
  s = fmaf(s, d2, 8.3333333e-3f);
  s = fmaf(s, d2, -1.6666667e-1f);
  *sn = fmaf(s * d2, d, d);
  float c = fmaf(d2, 2.4801587e-5f, -1.3888889e-3f);
  c = fmaf(c, d2, 4.1666667e-2f);
  c = fmaf(c, d2, -0.5f);
  *cs = fmaf(c, d2, 1.0f);
}

__global__ void __launch_bounds__(64) rdsp_sam_kernel(RdspSamParams p) {
  const size_t ch = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (ch >= (size_t)p.n_channels) return;
  const uint32_t gi = p.group_of ? (uint32_t)p.group_of[ch] : 0u;
  if (p.groups[gi].demod != RDSP_K_DEMOD_SAM) return;
  float omega = p.st_sam[ch * 4 + 1], fil = p.st_sam[ch * 4 + 2], dc = p.st_sam[ch * 4 + 3];
  float cs, sn;
  sincosf(p.st_sam[ch * 4 + 0], &sn, &cs);
  float *mi = p.mid + ch * p.mid_stride;
  const float *mq = p.mid_q + ch * p.mid_stride;
  float4 i4 = *reinterpret_cast<const float4 *>(mi), q4 = *reinterpret_cast<const float4 *>(mq);
#pragma unroll 1
  for (int n = 0; n < p.n_samples; n += 4) {
    const float I[4] = {i4.x, i4.y, i4.z, i4.w}, Q[4] = {q4.x, q4.y, q4.z, q4.w};
    if (n + 4 < p.n_samples) { /* the next quad lands while this one is in the loop */
      i4 = *reinterpret_cast<const float4 *>(mi + n + 4);
      q4 = *reinterpret_cast<const float4 *>(mq + n + 4);
    }
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      /* the oscillator for the next sample first: phs += fil with fil from the previous
       * sample, so this rotation does not wait for this sample's detector */
      float sd, cd;
      sam_sincos_small(fil, &sd, &cd);
      float c1 = cs * cd - sn * sd, s1 = sn * cd + cs * sd;
      if (k == 3) { /* one Newton step back to |.| = 1, once per quad (drift per rotation ~6e-8) */
        const float nrm = fmaf(-0.5f, c1 * c1 + s1 * s1, 1.5f);
        c1 *= nrm;
        s1 *= nrm;
      }
      const float corr0 = I[k] * cs + Q[k] * sn;
      const float corr1 = Q[k] * cs - I[k] * sn;
      const float mag2 = corr0 * corr0 + corr1 * corr1;
      const float det = sam_atan2(corr1, corr0) * (mag2 * __builtin_amdgcn_rcpf(mag2 + 1e-6f));
      omega = omega + p.g2 * det;
      omega = fminf(fmaxf(omega, p.wmin), p.wmax);
      fil = p.g1 * det + omega;
      cs = c1;
      sn = s1;
      dc = dc + (corr0 - dc) * (1.0f / 512.0f);
      o[k] = corr0 - dc;
    }
    *reinterpret_cast<float4 *>(mi + n) = make_float4(o[0], o[1], o[2], o[3]);
  }
  float phs = atan2f(sn, cs);
  if (phs < 0.0f) phs += 6.28318530717958647692f;
  p.st_sam[ch * 4 + 0] = phs;
  p.st_sam[ch * 4 + 1] = omega;
  p.st_sam[ch * 4 + 2] = fil;
  p.st_sam[ch * 4 + 3] = dc;
}

}  // namespace

extern "C" int rdsp_launch_sam(const RdspSamParams *p, hipStream_t stream) {
  if (p->n_samples <= 0 || (p->n_samples & 3) != 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rdsp_sam_kernel, dim3((p->n_channels + 63) / 64), dim3(64), 0, stream, *p);
  return (int)hipGetLastError();
}
