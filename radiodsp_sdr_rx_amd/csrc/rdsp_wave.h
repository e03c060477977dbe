/*
 * rdsp_wave.h -- wave-level helpers shared by the HIP kernels (device only).
 */
#ifndef RDSP_WAVE_H
#define RDSP_WAVE_H

#include "rdsp_front.h"
#include "rdsp_sync.h"

namespace {

/* A pointer that went through an `asm volatile("" : "+s"(p))` (to keep a table's loads inside a loop)
 * has lost its provenance: the compiler can no longer prove it global and emits FLAT loads, which
 * count on lgkmcnt as well as vmcnt -- every `s_waitcnt lgkmcnt(0)` of an LDS exchange behind them
 * then waits out the table's L2 round trip.  Saying "global" again gives `global_load ... s[base]`. */
struct GlobalF2 {
  typedef float v2 __attribute__((ext_vector_type(2)));
  const __attribute__((address_space(1))) v2 *p;
  __device__ __forceinline__ float2 operator[](int i) const {
    const v2 v = p[i];
    return make_float2(v[0], v[1]);
  }
};
__device__ __forceinline__ GlobalF2 as_global(const float2 *p) {
  return GlobalF2{(const __attribute__((address_space(1))) GlobalF2::v2 *)p};
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
/* sum over each 16-lane DPP row, result in every lane of the row */
__device__ __forceinline__ float row_allsum(float v) {
  v += dpp_f<0xB1>(v);  /* quad_perm [1,0,3,2] */
  v += dpp_f<0x4E>(v);  /* quad_perm [2,3,0,1] */
  v += dpp_f<0x141>(v); /* row_half_mirror */
  v += dpp_f<0x140>(v); /* row_mirror */
  return v;
}
/* sum over the 64-lane wave, wave-uniform result: the row sums, then row_bcast:15 into rows 1 and 3
 * and row_bcast:31 into rows 2 and 3 leave (r2 + r3) + (r0 + r1) in the last row -- the same pairs
 * as four v_readlane and three adds, to the bit, in 7 vector instructions instead of 13 */
__device__ __forceinline__ float wave_sum(float v) {
  v = row_allsum(v);
  /* masked adds (rows outside row_mask keep their value): the compiler's DPP combiner does not fold a
   * row-masked update_dpp into the add, so they are written out, wait states for the DPP reads included */
  /* ... and the wait state a v_readlane needs behind the VALU write of its source: the hazard recognizer
   * does not look into inline asm for what it wrote last, so the statement ends with its own s_nop */
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 0"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

/* `float_buffer_L[i] * 1.1` of CONV:334: the literal is a double, so the product is formed in
 * double and rounded to float once (x * 1.1f can differ in the last bit) */
__device__ __forceinline__ float mul_1p1(float x) { return (float)((double)x * 1.1); }

__device__ __forceinline__ float2 unpack_iq(uint32_t w, float si, float sq) {
  /* arm_q15_to_float: q/32768 (the 2^-15 is folded into si/sq, exact) */
  float xr = (float)(int16_t)(w & 0xFFFFu);
  float xi = (float)(int16_t)(w >> 16);
  return make_float2(xr * si, xi * sq);
}

/* arm_float_to_q15 of both channels into one word (CONV:346-347), in the variant the reference's firmware image holds
 * (CMSIS under ARM_MATH_ROUNDING): in = x * 32768; in += in > 0 ? 0.5f : -0.5f; (q15_t)__SSAT((q31_t)in, 16) -- round to
 * nearest, halves away from zero.  copysign(0.5, in) stands for the select (it differs for in = +0 only, where 0.5 and
 * -0.5 both convert to 0): v_bfi_b32 + v_add_f32; v_cvt_i32_f32 truncates (and saturates at int32, NaN -> 0, like
 * VCVT.S32.F32), v_cvt_pk_i16_i32 saturates to int16 and packs. */
__device__ __forceinline__ float q15_round_arg(float x) {
  const float v = x * 32768.0f;
  const float h = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, v) & 0x80000000u) | 0x3F000000u);
  return v + h;
}
__device__ __forceinline__ uint32_t pack_lr(float l, float r) {
  int a, b;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(a) : "v"(q15_round_arg(l)));
  asm("v_cvt_i32_f32 %0, %1" : "=v"(b) : "v"(q15_round_arg(r)));
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pk_i16(a, b));
}

}  // namespace

#endif
