/*
 * rdsp_q15.h -- the fixed-point radix-4 butterfly shared by the two analysers
 * (rdsp_spectrum.hip: AudioAnalyzeFFT256IQ, analyze_fft256iq.cpp:82; rdsp_fft1024.hip:
 * AudioAnalyzeFFT1024) and their integer square root (analyze_fft256iq.cpp:105).
 *
 * Arithmetic: arm_cfft_radix4_q15 as CMSIS-DSP publishes it for cores with the DSP extension (the
 * Cortex-M7 of the reference): two int16 per 32-bit word, real part in the low half, and per
 * butterfly, with a, b, c, d the inputs at distance N/4 (first stage: each >> 2 first),
 *   first stage   y0 = (sat(a+c) + sat(b+d)) >> 1          y2 = W2 . sat(sat(a+c) - sat(b+d))
 *                 y1 = W1 . sat((a-c) - j(b-d))             y3 = W3 . sat((a-c) + j(b-d))
 *   middle stages y0 = ((sat(a+c) + sat(b+d)) >> 1) >> 1    y2 = W2 . ((sat(a+c) - sat(b+d)) >> 1)
 *                 y1 = W1 . (((a-c) - j(b-d)) >> 1)         y3 = W3 . (((a-c) + j(b-d)) >> 1)
 *   last stage    the same sums >> 1, no twiddles
 * where W . y = ((cos y.re + sin y.im) >> 16, (cos y.im - sin y.re) >> 16) with (cos, sin) from
 * twiddleCoef_4096_q15 (floor(32768 x) clamped to int16; the reference's firmware image holds the
 * table, tests/test_firmware_tables.py), every sum of two saturated to int16 (__QADD16, __QSUB16,
 * __QASX, __QSAX) and every halving an arithmetic shift of the 17-bit sum (__SHADD16, __SHSUB16,
 * __SHASX, __SHSAX).  Total scale 1/N.  The kernels keep their own data flow (butterfly outputs
 * in DFT order k = 0..3 at positions base + k L, base-4 digit reversal at the end); CMSIS stores
 * y2 and y1 exchanged and reverses bits instead, which moves the same values to the same bins.
 *
 * gfx950 mapping: the saturating pair operations are v_pk_add_i16 / v_pk_sub_i16 with clamp, the
 * halving ones (a >> 1) + (b >> 1) + (a & b & 1) on packed halves, the exchange forms one half-word
 * swap plus a half-word select of the packed sum and difference, the twiddle product two
 * v_dot2_i32_i16 against (cos, sin) and (-sin, cos) and one v_perm_b32 that takes both high halves.
 */
#ifndef RDSP_Q15_H
#define RDSP_Q15_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rdsp_q15 {

typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s as_v(uint32_t w) { return __builtin_bit_cast(v2s, w); }
__device__ __forceinline__ uint32_t as_u(v2s v) { return __builtin_bit_cast(uint32_t, v); }

__device__ __forceinline__ int lo16(uint32_t w) { return (int)(int16_t)(w & 0xFFFFu); }
__device__ __forceinline__ int hi16(uint32_t w) { return (int)(int16_t)(w >> 16); }
__device__ __forceinline__ uint32_t pack16(int re, int im) { return ((uint32_t)re & 0xFFFFu) | ((uint32_t)im << 16); }

/* a.lo * b.lo + a.hi * b.hi in 32 bits (wraps like __SMUAD) */
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b) {
  int d;
  asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
/* (bits 16..31 of lo) | (bits 16..31 of hi) << 16 */
__device__ __forceinline__ uint32_t pack_high_halves(int lo, int hi) {
  return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x07060302u);
}
/* (x.lo, y.hi) */
__device__ __forceinline__ uint32_t lo_of_hi_of(uint32_t x, uint32_t y) {
  return __builtin_amdgcn_perm(y, x, 0x07060100u);
}
__device__ __forceinline__ uint32_t swap_halves(uint32_t x) { return __builtin_amdgcn_alignbit(x, x, 16); }

/* __QADD16, __QSUB16 */
__device__ __forceinline__ uint32_t qadd16(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_add_sat(as_v(a), as_v(b))); }
__device__ __forceinline__ uint32_t qsub16(uint32_t a, uint32_t b) { return as_u(__builtin_elementwise_sub_sat(as_v(a), as_v(b))); }
/* both halves >> n, arithmetic */
template <int N>
__device__ __forceinline__ uint32_t asr16(uint32_t a) { return as_u(as_v(a) >> (v2s)(short)N); }
/* __SHADD16: (a + b) >> 1 on the 17-bit sums; __SHSUB16 likewise */
__device__ __forceinline__ uint32_t shadd16(uint32_t a, uint32_t b) {
  return as_u(as_v(asr16<1>(a)) + as_v(asr16<1>(b)) + as_v(a & b & 0x00010001u));
}
__device__ __forceinline__ uint32_t shsub16(uint32_t a, uint32_t b) {
  return as_u(as_v(asr16<1>(a)) - as_v(asr16<1>(b)) - as_v(~a & b & 0x00010001u));
}

/* the twiddle W = (cos, sin) as the two operands of W . y */
struct Twiddle {
  uint32_t a; /* ( cos, sin): real part      cos y.re + sin y.im */
  uint32_t b; /* (-sin, cos): imaginary part cos y.im - sin y.re */
};
/* w = cos | sin << 16 (rdsp_q15_twiddles); sin > -32768 for every entry a 256- or 1024-point plan reads */
__device__ __forceinline__ Twiddle make_twiddle(uint32_t w) {
  return Twiddle{w, pack16(-hi16(w), lo16(w))};
}
/* out1 = __SMUAD(C, y) >> 16, out2 = __SMUSDX(C, y), word = (out2 & 0xFFFF0000) | (out1 & 0xFFFF) */
__device__ __forceinline__ uint32_t twiddle_mul(uint32_t y, const Twiddle &w) {
  return pack_high_halves(dot2(y, w.a), dot2(y, w.b));
}

enum { kFirstStage = 0, kMiddleStage = 1, kLastStage = 2 };

/* x = (a, b, c, d) -> (y0, y1, y2, y3); w[1..3] = W^j, W^2j, W^3j (unused in the last stage) */
template <int STAGE>
__device__ __forceinline__ void bfly(uint32_t *x, const Twiddle *w) {
  uint32_t a = x[0], b = x[1], c = x[2], d = x[3];
  if (STAGE == kFirstStage) {
    a = asr16<2>(a); b = asr16<2>(b); c = asr16<2>(c); d = asr16<2>(d);
  }
  const uint32_t R = qadd16(a, c), S = qsub16(a, c), V = qadd16(b, d);
  const uint32_t Tx = swap_halves(qsub16(b, d));
  /* with T = b - d: A = (S.lo + T.hi, S.hi + T.lo), B = (S.lo - T.hi, S.hi - T.lo);
   * (a-c) - j(b-d) = (A.lo, B.hi), (a-c) + j(b-d) = (B.lo, A.hi) */
  if (STAGE == kFirstStage) {
    const uint32_t A = qadd16(S, Tx), B = qsub16(S, Tx);
    x[0] = shadd16(R, V);
    x[2] = twiddle_mul(qsub16(R, V), w[2]);
    x[1] = twiddle_mul(lo_of_hi_of(A, B), w[1]);
    x[3] = twiddle_mul(lo_of_hi_of(B, A), w[3]);
  } else if (STAGE == kMiddleStage) {
    const uint32_t A = shadd16(S, Tx), B = shsub16(S, Tx);
    x[0] = asr16<1>(shadd16(R, V));
    x[2] = twiddle_mul(shsub16(R, V), w[2]);
    x[1] = twiddle_mul(lo_of_hi_of(A, B), w[1]);
    x[3] = twiddle_mul(lo_of_hi_of(B, A), w[3]);
  } else {
    const uint32_t A = shadd16(S, Tx), B = shsub16(S, Tx);
    x[0] = shadd16(R, V);
    x[2] = shsub16(R, V);
    x[1] = lo_of_hi_of(A, B);
    x[3] = lo_of_hi_of(B, A);
  }
}

/* multiply_16tx16t_add_16bx16b(w, w), analyze_fft256iq.cpp:89 */
__device__ __forceinline__ uint32_t magsq(uint32_t w) { return (uint32_t)dot2(w, w); }

/* q15 window on both components: (x * w) >> 15 = high half-word of x * 2w; w2 = 2 w */
__device__ __forceinline__ uint32_t window_mul(uint32_t x, int w2) {
  return pack_high_halves(__mul24(lo16(x), w2), __mul24(hi16(x), w2));
}

/* sqrt_uint32_approx of Teensy Audio's utility/sqrt_integer.h (analyze_fft256iq.cpp:105): first
 * guess from a 33-entry table by the count of leading zeros (`guess`: rdsp_sqrt_guess_table, the
 * reference's firmware image holds it), two integer Newton steps.  in = 0: guess 0, and the
 * Cortex-M7's UDIV by zero yields 0, so the result is 0. */
__device__ __forceinline__ uint32_t sqrt_uint32_approx(uint32_t in, const uint16_t *guess) {
  if (in == 0u) return 0u;
  uint32_t n = guess[__clz((int)in)];
  n = ((in / n) + n) >> 1;
  n = ((in / n) + n) >> 1;
  return n;
}

}  // namespace rdsp_q15
#endif
