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
 * to four outputs (register reuse 3.8x); every tap index is a compile-time
 * constant and wave-uniform, so taps are LDS broadcast reads of four at a time.
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
 * outputs m = 4l + r.  taps = [4][16] float4 in branch order (hc[c][k'] =
 * h[4k'+c]), read with wave-uniform addresses (LDS broadcast on the device): the
 * read for step d needs taps 4d-3 .. 4d+3, i.e. the float4 of step d-1 and d.
 * Data and tap reads of step d+1 are issued before the FMAs of step d, and all
 * of them are DS operations, so the waits are counted (lgkmcnt(N)), never drained. */
template <bool HAND = false>
RDSP_HD void fir_lane(int l, int c0, int c1, const float2 *xs, const float4 *taps, float2 *acc) {
  const float4 *xs4 = reinterpret_cast<const float4 *>(xs);
#ifdef __HIP_DEVICE_COMPILE__
  if constexpr (HAND) {
  /* HAND: the packed FMAs written out -- tap = one half of an aligned pair of the tap window,
   * broadcast by op_sel, times a complex sample -- with the reads two steps ahead of the FMAs
   * that use them.  The compiler's own code for the loop below copies 17 taps per branch into
   * fresh pairs (v_mov) and holds 14-26 more VGPRs; without them the un-overlapped front kernel
   * is 2 % (FFT_L 512) to 7 % (4096) faster.  Not for FFT_L 256, where it measured 8 % slower
   * (10 % with the kernel held at two waves per SIMD by LDS padding, so not an occupancy effect;
   * unexplained), hence the template switch. */
  rdsp_v2f a[4];
#pragma unroll
  for (int r = 0; r < 4; r++) a[r] = rdsp_v2f{acc[r].x, acc[r].y};
  for (int c = c0; c < c1; c++) {
    const float4 *pl = xs4 + (c * 2) * RDSP_XP + l + 16;
    const float4 *tp = taps + c * 16;
    float4 X0n[2], X1n[2], Tn[2];
    X0n[0] = pl[0]; X1n[0] = pl[RDSP_XP]; Tn[0] = tp[0];
    X0n[1] = pl[-1]; X1n[1] = pl[RDSP_XP - 1]; Tn[1] = tp[1];
    float4 tA = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int d = 0; d <= 16; d++) {
      const int cur = d & 1;
      const float4 X0 = X0n[cur], X1 = X1n[cur], tB = Tn[cur];
      if (d + 2 <= 16) {
        X0n[cur] = pl[-(d + 2)];
        X1n[cur] = pl[RDSP_XP - (d + 2)];
        if (d + 2 <= 15) Tn[cur] = tp[d + 2];
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (d == 16 && q == 0) continue;
        const float4 Xq = (q < 2) ? X0 : X1;
        const rdsp_v2f xv = (q & 1) ? rdsp_v2f{Xq.z, Xq.w} : rdsp_v2f{Xq.x, Xq.y};
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kp = 4 * d + r - q;
          if (kp >= 0 && kp <= 63) {
            const int ti = 4 + r - q; /* index into the window {tA, tB} */
            const rdsp_v2f tp2 = (ti < 4) ? ((ti < 2) ? rdsp_v2f{tA.x, tA.y} : rdsp_v2f{tA.z, tA.w})
                                          : ((ti < 6) ? rdsp_v2f{tB.x, tB.y} : rdsp_v2f{tB.z, tB.w});
            if (ti & 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(a[r]) : "v"(tp2), "v"(xv));
            else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(a[r]) : "v"(tp2), "v"(xv));
          }
        }
      }
      tA = tB;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) acc[r] = make_float2(a[r].x, a[r].y);
  return;
  }
#endif
  for (int c = c0; c < c1; c++) {
    const float4 *pl = xs4 + (c * 2) * RDSP_XP + l + 16;
    const float4 *tp = taps + c * 16;
    float4 tA = make_float4(0.f, 0.f, 0.f, 0.f), tB = tp[0];
    float4 X0 = pl[0], X1 = pl[RDSP_XP];
#pragma unroll
    for (int d = 0; d <= 16; d++) {
      const float4 Xc[2] = {X0, X1};
      float4 tN = tB;
      if (d + 1 <= 16) {
        X0 = pl[-(d + 1)];
        X1 = pl[RDSP_XP - (d + 1)];
        if (d + 1 <= 15) tN = tp[d + 1];
      }
      const float ta[8] = {tA.x, tA.y, tA.z, tA.w, tB.x, tB.y, tB.z, tB.w}; /* taps 4d-4 .. 4d+3 */
#pragma unroll
      for (int qh = 0; qh < 2; qh++) {
#pragma unroll
        for (int ql = 0; ql < 2; ql++) {
          const int q = 2 * qh + ql;
          if (d == 16 && q == 0) continue;
          const float Xx = ql ? Xc[qh].z : Xc[qh].x, Xy = ql ? Xc[qh].w : Xc[qh].y;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int kp = 4 * d + r - q;
            if (kp >= 0 && kp <= 63) {
              const float t = ta[4 + r - q];
              acc[r].x = fmaf(t, Xx, acc[r].x);
              acc[r].y = fmaf(t, Xy, acc[r].y);
            }
          }
        }
      }
      tA = tB;
      tB = tN;
#ifdef __HIP_DEVICE_COMPILE__
      /* one step = 3 LDS reads (for step d+1) + 16 packed FMAs: keep that shape so
       * the scheduler does not hoist a whole branch of reads into registers */
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
#endif
    }
  }
}

/* ---- the same FIR on the matrix pipe ------------------------------------------------
 * With m = 16a + b:  y[16a+b] = sum_d x[64a+d] h[4b-d],  d in [-256, 63], i.e. a GEMM
 *     Y[a][b] = A[a][d] B[d][b],   A[a][d] = x[64a + d]  (the chunk and its history, reshaped)
 *                                   B[d][b] = h[4b - d]   (banded Toeplitz of the taps)
 * of 16 x 320 x 16 per component: 80 K-slices of v_mfma_f32_16x16x4_f32, 320/256 of the
 * direct form's multiplies but on the matrix pipe, which the rest of the chain leaves idle
 * (the chain is fp32-VALU-bound).  A comes from LDS with one ds_read_b64 (re, im) per lane
 * and slice: the input is stored linearly with two float2 of padding per 64 samples so the
 * 16 rows of a slice fall in different banks; B is one ds_read_b32 per lane and slice
 * from the tap line hz[t + 64] = h[t], zero outside 0..255. */
constexpr int RDSP_XL_N = 1280 + (1280 / 64) * 2; /* float2: 256 history + 1024 chunk, padded */
constexpr int RDSP_HZ_N = 384;                    /* floats: taps -64 .. 319 */
/* float2 index of chunk-local input sample n, n in [-256, 1023] */
constexpr RDSP_HD int xl_pos(int n) { return (n + 256) + ((n + 256) >> 6) * 2; }
typedef float rdsp_v4f __attribute__((ext_vector_type(4)));
/* K-slices [s0, s1) of the product; lane l = 16 kq + i holds, in d?[r], output m = 64 kq + 16 r + i */
template <int S0, int S1>
__device__ __forceinline__ void fir_matrix(int lane, const float2 *xl, const float *hz, rdsp_v4f &dre, rdsp_v4f &dim) {
#ifdef __HIP_DEVICE_COMPILE__
  const int i = lane & 15, kq = lane >> 4;
  const float2 *ap = xl + (66 * i + kq); /* xl_pos(64 i + kq - 256) */
  const float *bp = hz + (320 + 4 * i - kq);
#pragma unroll
  for (int s = S0; s < S1; s++) {
    const float2 a = ap[4 * s + (s >> 4) * 2]; /* 4 s + kq never crosses a 64-sample row on its own */
    const float b = bp[-4 * s];
    dre = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, dre, 0, 0, 0);
    dim = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, dim, 0, 0, 0);
  }
#else
  (void)lane; (void)xl; (void)hz; (void)dre; (void)dim;
#endif
}

/* phasor of a 32-bit phase by the ALU: exp(-j*2*pi*ph/2^32).  The top 24 bits go
 * through sincospi (exact argument), the low 8 bits are a first-order residual
 * (angle < 3.7e-7 rad). */
RDSP_HD float2 nco_phasor_alu(uint32_t ph) {
  const float xh = (float)(ph >> 8) * 1.1920928955078125e-07f; /* (ph>>8) / 2^23 = turns*2 */
  const float r = (float)(ph & 255u) * 1.4629180792671596e-09f; /* 2*pi/2^32 */
  float sn, cs;
#ifdef __HIP_DEVICE_COMPILE__
  sincospif(xh, &sn, &cs);
#else
  sn = (float)sin(3.14159265358979323846 * (double)xh);
  cs = (float)cos(3.14159265358979323846 * (double)xh);
#endif
  /* (cs - j sn)(1 - j r) */
  return make_float2(fmaf(-sn, r, cs), -fmaf(cs, r, sn));
}

/* the mixer's complex product: its phasors must come out bit-identical wherever they are
 * recomputed (FIR history at the start of a call).  cmul() is two hand-placed packed
 * instructions on the device, re = fma(a.x, b.x, -(a.y b.y)), im = fma(a.x, b.y, a.y b.x),
 * so no compiler contraction choice can differ between the two places. */
RDSP_HD float2 cmul_pinned(float2 a, float2 b) { return cmul(a, b); }
/* ... with b a field of the group record (wave-uniform) */
RDSP_HD float2 cmul_pinned_u(float2 a, float2 b) { return cmul_uniform(a, b); }

/* arm_float_to_q15 semantics (CONV:346-347; the ARM_MATH_ROUNDING variant, which is what the reference's firmware
 * image holds): x*32768, +-0.5 by sign, truncate, saturate */
RDSP_HD int q15_of_float(float x) {
  float v = x * 32768.0f;
  v = v + (v > 0.0f ? 0.5f : -0.5f);
  v = fminf(fmaxf(v, -32768.0f), 32767.0f);   /* (NaN -> -32768 here where the conversion instruction gives 0: rdsp_float_to_q15 documents finite input) */
  return (int)v;
}

}  // namespace rdsp

#endif
