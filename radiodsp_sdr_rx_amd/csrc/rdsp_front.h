/*
 * rdsp_front.h -- host/device helpers of the front kernel: polyphase LDS layout
 * of the mixed input stream, the decimating FIR lane function and the NCO
 * phasor.  __host__ __device__ so tests/host/host_fir_check.cpp runs the same
 * index arithmetic on the CPU.
 *
 * Decimator (build-defined stage A3; the reference has none):
 *     y[m] = sum_{k<256} h[k] * x[4m - k]
 * With k = 4k' + c:  y[m] = sum_c sum_{k'<64} hc[c][k'] * X_c[m - k'],
 * X_c[u] = x[4u - c].  One chunk = 1024 input samples -> 256 outputs.  Lane l
 * of a wave owns outputs m = 4l..4l+3, so each X_c[u] it loads from LDS feeds up
 * to four outputs (register reuse 3.8x) and every tap index is a compile-time
 * constant -> taps come through scalar loads, not LDS or VGPRs.
 *
 * LDS layout: X_c[u], u = 4v + q, lives in plane (c, q>>1) at entry v + 16
 * (v in [-16, 64]), two q per 16-byte entry.  Lanes read entry (l - d + 16) of
 * one plane: consecutive 16-byte words -> conflict-free ds_read_b128.
 */
#ifndef RDSP_FRONT_H
#define RDSP_FRONT_H

#include "rdsp_fft.h"
#include "rdsp_kernels.h"

namespace rdsp {

/* float2 index of X_c[u] (u = 4v + q) inside the plane array.  Planes are
 * float4-wide: plane (c, q>>1) entry e = v + 16 holds {X(q even), X(q odd)}, so
 * a lane fetches two of its inputs with one aligned ds_read_b128 and lanes read
 * consecutive 16-byte words (conflict-free, full LDS rate). */
RDSP_HD int xs_off(int c, int q, int e) { return ((c * 2 + (q >> 1)) * RDSP_XP + e) * 2 + (q & 1); }

/* LDS slot (float2 index) of chunk-local input sample n, n in [-256, 1023] */
RDSP_HD int xs_pos(int n) {
  int u = (n + 3) >> 2; /* ceil(n/4), arithmetic shift */
  int c = 4 * u - n;
  return xs_off(c, u & 3, (u >> 2) + 16);
}

/* partial FIR of lane l over polyphase branches [c0, c1): acc[r] += ... for the
 * outputs m = 4l + r.  hc = [4][64] taps in branch order. */
RDSP_HD void fir_lane(int l, int c0, int c1, const float2 *xs, const float *hc, float2 *acc) {
  const float4 *xs4 = reinterpret_cast<const float4 *>(xs);
  for (int c = c0; c < c1; c++) {
    /* all 64 taps of the branch go to SGPRs up front (one scalar-load wait per
     * branch); inside the d-loop only LDS reads use lgkmcnt, so the compiler can
     * count them instead of draining to zero after every read */
    float h[64];
#pragma unroll
    for (int k = 0; k < 64; k++) h[k] = hc[c * 64 + k];
#ifdef __HIP_DEVICE_COMPILE__
    __builtin_amdgcn_sched_barrier(0);
#endif
    const float4 *pl = xs4 + (c * 2) * RDSP_XP + l + 16;
#pragma unroll
    for (int d = 0; d <= 16; d++) {
#pragma unroll
      for (int qh = 0; qh < 2; qh++) {
        float4 X4 = pl[qh * RDSP_XP - d];
#pragma unroll
        for (int ql = 0; ql < 2; ql++) {
          const int q = 2 * qh + ql;
          if (d == 16 && q == 0) continue;
          const float Xx = ql ? X4.z : X4.x, Xy = ql ? X4.w : X4.y;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int kp = 4 * d + r - q;
            if (kp >= 0 && kp <= 63) {
              float t = h[kp];
              acc[r].x = fmaf(t, Xx, acc[r].x);
              acc[r].y = fmaf(t, Xy, acc[r].y);
            }
          }
        }
#ifdef __HIP_DEVICE_COMPILE__
        /* keep one LDS read in flight per 8 packed FMAs instead of letting the
         * scheduler hoist all 34 reads (136 VGPRs) to the top of the branch */
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#endif
      }
    }
  }
}

/* exp(-j*2*pi*ph/2^32) from two 1024-entry tables and a first-order residual:
 * ph = i1*2^22 + i2*2^12 + rem; the residual angle is < 2*pi/2^20, so
 * exp(-j r) = 1 - j r to 2e-11. */
RDSP_HD float2 nco_phasor(uint32_t ph, const float2 *t1, const float2 *t2) {
  float2 a = t1[ph >> 22];
  float2 b = t2[(ph >> 12) & 1023u];
  float r = (float)(ph & 4095u) * 1.4629180792671596e-09f; /* 2*pi/2^32 */
  float2 p = cmul(a, b);
  return make_float2(fmaf(p.y, r, p.x), fmaf(-p.x, r, p.y));
}

/* arm_float_to_q15 semantics (CONV:346-347): x*32768, truncate, saturate */
RDSP_HD int q15_of_float(float x) {
  float v = x * 32768.0f;
  v = fminf(fmaxf(v, -32768.0f), 32767.0f);
  return (int)v;
}

}  // namespace rdsp

#endif
