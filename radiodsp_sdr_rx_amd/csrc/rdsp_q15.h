/*
 * rdsp_q15.h -- the fixed-point radix-4 butterfly shared by the two analysers
 * (rdsp_spectrum.hip: AudioAnalyzeFFT256IQ, analyze_fft256iq.cpp:82; rdsp_fft1024.hip:
 * AudioAnalyzeFFT1024).  Arithmetic as the test restatement defines it (arm_cfft_radix4_q15 role):
 *   y_k = (four-term sum) >> 2,   out_k = sat16((y_k * W_k) >> 15)   on both components,
 * every intermediate exact in 32 bits.
 *
 * gfx950 mapping.  A complex value is one register, re in the low and im in the high half-word.
 * The first level of sums reads the halves with sign extension through SDWA operand selects (the
 * compiler emits those from the casts), the products are two v_dot2_i32_i16 against the twiddle
 * kept twice, as (wr, -wi) and (wi, wr) -- the table holds round(32767 cos), round(-32767 sin), so
 * -wi always fits -- and v_cvt_pk_i16_i32 saturates and packs both components at once.
 * W^0 = (32767, 0) has no cross terms and cannot saturate: (y * 32767) >> 15 is the high half-word
 * of y * 65534, picked out of the two products by one v_perm_b32.
 */
#ifndef RDSP_Q15_H
#define RDSP_Q15_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdsp_q15 {

__device__ __forceinline__ int lo16(uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ int hi16(uint32_t w) { return (int)(int16_t)(w >> 16); }
__device__ __forceinline__ uint32_t pack16(int re, int im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }

/* a.lo * b.lo + a.hi * b.hi, exact (|result| < 2^31 for q15 operands) */
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
/* saturate both to int16 and pack (lo, hi) */
__device__ __forceinline__ uint32_t sat_pack(int lo, int hi) {
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(lo, hi));
}
/* (bits 16..31 of lo) | (bits 16..31 of hi) << 16 */
__device__ __forceinline__ uint32_t pack_high_halves(int lo, int hi) {
  return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060302u);
}

/* the twiddle W = (wr, wi) as the two operands of the complex product */
struct Twiddle {
  uint32_t a; /* (wr, -wi): real part      y.re * wr - y.im * wi */
  uint32_t b; /* (wi,  wr): imaginary part y.re * wi + y.im * wr */
};
__device__ __forceinline__ Twiddle make_twiddle(uint32_t w) {
  const int wr = lo16(w), wi = hi16(w);
  return Twiddle{pack16(wr, -wi), pack16(wi, wr)};
}

/* the four sums of a radix-4 butterfly before the >> 2 (18-bit values) */
struct Sums { int r[4], i[4]; };
__device__ __forceinline__ Sums bfly_sums(const uint32_t *x) {
  const int ar = lo16(x[0]), ai = hi16(x[0]), br = lo16(x[1]), bi = hi16(x[1]);
  const int cr = lo16(x[2]), ci = hi16(x[2]), dr = lo16(x[3]), di = hi16(x[3]);
  const int s0r = ar + cr, s0i = ai + ci, s1r = ar - cr, s1i = ai - ci;
  const int s2r = br + dr, s2i = bi + di, s3r = br - dr, s3i = bi - di;
  Sums s;
  s.r[0] = s0r + s2r; s.i[0] = s0i + s2i;
  s.r[1] = s1r + s3i; s.i[1] = s1i - s3r;
  s.r[2] = s0r - s2r; s.i[2] = s0i - s2i;
  s.r[3] = s1r - s3i; s.i[3] = s1i + s3r;
  return s;
}

/* one output of a butterfly times a general twiddle */
__device__ __forceinline__ uint32_t twiddle_mul(int sr, int si, const Twiddle &w) {
  const uint32_t y = sat_pack(sr >> 2, si >> 2); /* both fit: the pack is exact */
  return sat_pack(dot2(y, w.a) >> 15, dot2(y, w.b) >> 15);
}
/* ... times W^0 = (32767, 0) */
__device__ __forceinline__ uint32_t twiddle_mul_w0(int sr, int si) {
  return pack_high_halves(__mul24(sr >> 2, 65534), __mul24(si >> 2, 65534));
}

/* x[k] (packed) -> sat16(((sum_k) >> 2) * W_k >> 15); w[0] is W^0 in every stage of a
 * decimation-in-frequency pass (k j = 0), w[1..3] are general */
__device__ __forceinline__ void bfly(uint32_t *x, const Twiddle *w) {
  const Sums s = bfly_sums(x);
  x[0] = twiddle_mul_w0(s.r[0], s.i[0]);
#pragma unroll
  for (int k = 1; k < 4; k++) x[k] = twiddle_mul(s.r[k], s.i[k], w[k]);
}
/* last stage: every twiddle is W^0 */
__device__ __forceinline__ void bfly_w0(uint32_t *x) {
  const Sums s = bfly_sums(x);
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = twiddle_mul_w0(s.r[k], s.i[k]);
}
/* last stage, components left unpacked (they feed re^2 + im^2) */
__device__ __forceinline__ void bfly_w0_unpacked(const uint32_t *x, int *re, int *im) {
  const Sums s = bfly_sums(x);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    re[k] = __mul24(s.r[k] >> 2, 65534) >> 16;
    im[k] = __mul24(s.i[k] >> 2, 65534) >> 16;
  }
}

/* q15 window on both components: (x * w) >> 15 = high half-word of x * 2w; w2 = 2 w */
__device__ __forceinline__ uint32_t window_mul(uint32_t x, int w2) {
  return pack_high_halves(__mul24(lo16(x), w2), __mul24(hi16(x), w2));
}

__device__ __forceinline__ uint32_t isqrt32(uint32_t x) {
  uint32_t r = (uint32_t)sqrtf((float)x);
  while ((unsigned long long)r * r > x) r--;
  while ((unsigned long long)(r + 1) * (r + 1) <= x) r++;
  return r;
}

}  // namespace rdsp_q15
#endif
