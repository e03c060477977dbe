/*
 * rdsp_oracle.c -- CPU ORACLE (test infrastructure; see rdsp_oracle.h header).
 * PARITY: PINNED ON THE REFERENCE'S OWN COMPILED CODE for what its shipped build contains -- routines of the firmware
 * image (the .hex under pre_compiled) are run under an instruction-set interpreter in the build container and their inputs and
 * outputs kept as tests/golden/firmware_kat.npz (tests/test_firmware_kat.py): arm_cfft_radix4_q15, arm_lms_norm_f32,
 * arm_biquad_cascade_df1_f32, the q15 converters, arm_cmplx_mult_cmplx_f32, both analysers' update() and
 * AudioFilterBiquad's BIT FOR BIT; the design routine's taps identical once narrowed to float; the CONV stage
 * (doConvolutionalProcessing with and without the NLMS, level and pass-band changes in mid-stream, the filter-off branch
 * as written) 1.6e-7 ... 2.0e-6 normwise, int16 within one count.  PINNED BY REFERENCE-HELD DATA besides -- the constant
 * tables of the same image (tests/golden/firmware_tables.npz, tests/test_firmware_tables.py).  UNPINNED for the stages
 * that are not in the image or not in the tree: the decimator (not in the reference), the spectral stage (src/backup
 * only), and the build-defined stand-ins for the AudioSDR engine (NCO, ALS, AGC, SAM, blanker) -- those are anchored by
 * analytic KATs and an independent float64 model only.
 *
 * Citation short names (relative to /root/reference/src):
 *   CONV = RadioDSP_SDR_RX/RDSP_convolutional.h
 *   NR   = RadioDSP_SDR_RX/RDSP_noise_reduction.h
 *   SPEC = backup/RDSP_convolutional_spec.h
 *   INO  = RadioDSP_SDR_RX/RadioDSP_SDR_RX.ino
 *
 * Arithmetic types mirror the reference: float32_t -> float, double -> double.
 * Build with -ffp-contract=off so no FMA contraction changes the rounding.
 */
#include "rdsp_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_PI 3.14159265358979323846
#define ORC_TWO_PI (2.0 * ORC_PI)
#define LMS_TAPS 96   /* MAX_LMS_TAPS, NR:23; calc_taps NR:39 */
#define LMS_DELAY 128 /* MAX_LMS_DELAY, NR:24 */

/* ------------------------------------------------------------------------ */
/* CMSIS-DSP primitive restatements (published semantics)        */
/* ------------------------------------------------------------------------ */

/* arm_q15_to_float: dst = (float)src / 32768.0f   (call sites CONV:241-242) */
void orc_q15_to_float(const int16_t *src, float *dst, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) dst[i] = (float)src[i] / 32768.0f;
}

/* arm_float_to_q15 (call sites CONV:346-347) in the variant the reference's firmware image holds -- the one CMSIS
 * compiles under ARM_MATH_ROUNDING:
 *     in = *pIn++ * 32768.0f;  in += in > 0.0f ? 0.5f : -0.5f;  *pDst++ = (q15_t)__SSAT((q31_t)in, 16);
 * i.e. round to nearest, halves away from zero (the image: VMOV #0.5 / #-0.5, VMUL, VCMP #0, VADD, VCVT toward zero,
 * SSAT #16 per sample; tests/test_firmware_tables.py).  Rounds 1-4 restated the other variant (plain truncation).
 * The cast saturates like VCVT.S32.F32 does, NaN gives 0. */
void orc_float_to_q15(const float *src, int16_t *dst, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    float v = src[i] * 32768.0f;
    v += v > 0.0f ? 0.5f : -0.5f;
    int32_t q;
    if (v != v) q = 0;
    else if (v <= -2147483648.0f) q = INT32_MIN;
    else if (v >= 2147483648.0f) q = INT32_MAX;
    else q = (int32_t)v;
    if (q > 32767) q = 32767;
    if (q < -32768) q = -32768;
    dst[i] = (int16_t)q;
  }
}

/* arm_cfft_f32(S, p, ifftFlag, bitReverseFlag=1): in-place interleaved complex
 * float transform, natural order in and out, inverse scaled by 1/N.  CMSIS
 * uses a radix-8/4 decomposition; this is an iterative radix-2 DIT with float
 * twiddles (rounded from double).  Agreement with a float64 DFT is ~1e-7
 * normwise (tests/test_oracle_kat.py). */
void orc_cfft_f32(float *buf, uint32_t n, int inverse) {
  /* bit reversal */
  for (uint32_t i = 1, j = 0; i < n; i++) {
    uint32_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      float tr = buf[2 * i], ti = buf[2 * i + 1];
      buf[2 * i] = buf[2 * j];
      buf[2 * i + 1] = buf[2 * j + 1];
      buf[2 * j] = tr;
      buf[2 * j + 1] = ti;
    }
  }
  for (uint32_t len = 2; len <= n; len <<= 1) {
    uint32_t half = len >> 1;
    for (uint32_t k = 0; k < half; k++) {
      double ang = -ORC_TWO_PI * (double)k / (double)len;
      float wr = (float)cos(ang);
      float wi = (float)sin(ang);
      if (inverse) wi = -wi;
      for (uint32_t s = k; s < n; s += len) {
        uint32_t a = s, b = s + half;
        float xr = buf[2 * b] * wr - buf[2 * b + 1] * wi;
        float xi = buf[2 * b] * wi + buf[2 * b + 1] * wr;
        buf[2 * b] = buf[2 * a] - xr;
        buf[2 * b + 1] = buf[2 * a + 1] - xi;
        buf[2 * a] = buf[2 * a] + xr;
        buf[2 * a + 1] = buf[2 * a + 1] + xi;
      }
    }
  }
  if (inverse) {
    float sc = 1.0f / (float)n;
    for (uint32_t i = 0; i < 2 * n; i++) buf[i] *= sc;
  }
}

/* arm_cmplx_mult_cmplx_f32 (call site CONV:301) */
void orc_cmplx_mult_cmplx_f32(const float *a, const float *b, float *dst, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    float ar = a[2 * i], ai = a[2 * i + 1], br = b[2 * i], bi = b[2 * i + 1];
    dst[2 * i] = ar * br - ai * bi;
    dst[2 * i + 1] = ar * bi + ai * br;
  }
}

/* arm_cmplx_mag_f32 (call site SPEC:182) */
void orc_cmplx_mag_f32(const float *src, float *dst, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    float re = src[2 * i], im = src[2 * i + 1];
    dst[i] = sqrtf(re * re + im * im);
  }
}

/* ------------------------------------------------------------------------ */
/* CONV:127-185  windowed-sinc complex band-pass design (double precision)   */
/* ------------------------------------------------------------------------ */
static double window_value(int window, int i, int n) {
  double den = (double)(n - 1);
  switch (window) {
    case 1: /* 4-term Blackman-Harris, CONV:153-158 */
      return 0.35875 - 0.48829 * cos((ORC_TWO_PI * i) / den) +
             0.14128 * cos((2.0 * ORC_TWO_PI * i) / den) -
             0.01168 * cos((3.0 * ORC_TWO_PI * i) / den);
    case 2: /* CONV:159-164 */
      return 0.355768 - 0.487396 * cos((ORC_TWO_PI * i) / den) +
             0.144232 * cos((2.0 * ORC_TWO_PI * i) / den) -
             0.012604 * cos((3.0 * ORC_TWO_PI * i) / den);
    case 3: /* cosine, CONV:165-168 ((float32_t)i cast is exact for i < 2^24) */
      return cos((ORC_PI * (double)(float)i) / den);
    case 4: /* Hann, CONV:169-172 */
      return 0.5 * (1.0 - cos(ORC_PI * 2 * (double)i / den));
    default: /* Blackman-Nuttall, CONV:173-179 */
      return 0.3635819 - 0.4891775 * cos((ORC_TWO_PI * i) / den) +
             0.1365995 * cos((2.0 * ORC_TWO_PI * i) / den) -
             0.0106411 * cos((3.0 * ORC_TWO_PI * i) / den);
  }
}

void orc_calc_cplx_FIR_coeffs(double *coeffs_I, double *coeffs_Q, int numCoeffs,
                              double FLoCut, double FHiCut, double SampleRate,
                              int window) {
  double nFL = FLoCut / SampleRate;               /* CONV:132 */
  double nFH = FHiCut / SampleRate;               /* CONV:133 */
  double nFc = (nFH - nFL) / 2.0;                 /* CONV:134 */
  double nFs = ORC_PI * (nFH + nFL);              /* CONV:135 */
  double fCenter = 0.5 * (double)(numCoeffs - 1); /* CONV:136 */
  for (int i = 0; i < numCoeffs; i++) {
    double x = (double)(float)i - fCenter; /* CONV:147 */
    double z;
    if (fabs((double)i - fCenter) < 0.01) /* CONV:149-150 */
      z = 2.0 * nFc;
    else
      z = sin(ORC_TWO_PI * x * nFc) / (ORC_PI * x) * window_value(window, i, numCoeffs);
    coeffs_I[i] = z * cos(nFs * x); /* CONV:182 */
    coeffs_Q[i] = z * sin(nFs * x); /* CONV:183 */
  }
}

/* CONV:87-110.  Taps narrowed to float on store (CONV:98-99); zero fill starts
 * at index FFT_length+1 (CONV:102) which overwrites Q[FFT_L/2] -- reproduced. */
void orc_init_filter_mask(float *mask, const double *coef_I, const double *coef_Q,
                          uint32_t fft_l) {
  uint32_t ntaps = fft_l / 2 + 1; /* m_NumTaps, CONV:72 */
  for (uint32_t i = 0; i < ntaps; i++) {
    mask[2 * i] = (float)coef_I[i];
    mask[2 * i + 1] = (float)coef_Q[i];
  }
  for (uint32_t i = fft_l + 1; i < 2 * fft_l; i++) mask[i] = 0.0f;
  orc_cfft_f32(mask, fft_l, 0); /* CONV:108 */
}

/* ------------------------------------------------------------------------ */
/* NLMS instance: arm_lms_norm_instance_f32 + the reference's delay ring     */
/* ------------------------------------------------------------------------ */
typedef struct {
  float mu;
  float energy, x0;
  float coeffs[LMS_TAPS + LMS_DELAY]; /* LMS_NormCoeff_f32, NR:31 */
  float state[LMS_TAPS + LMS_DELAY];  /* LMS_StateF32, NR:30      */
  float delay[256 + LMS_DELAY];       /* LMS_nr_delay, NR:32      */
  float errsig[256 + 10];             /* LMS_errsig1, NR:26       */
  uint32_t inbuf, outbuf;             /* statics, NR:69           */
} orc_lms_t;

/* NR:35-64 */
static void lms_init(orc_lms_t *s, int strength) {
  float mu_calc = (float)strength; /* NR:48 */
  mu_calc /= 2;                    /* NR:51 */
  mu_calc += 2;                    /* NR:52 */
  mu_calc /= 10;                   /* NR:53 */
  mu_calc = powf(10, mu_calc);     /* NR:54 */
  mu_calc = 1 / mu_calc;           /* NR:55 */
  s->mu = mu_calc;
  for (int i = 0; i < 256 + 128; i++) s->delay[i] = 0.0f;      /* NR:58 */
  for (int i = 0; i < LMS_TAPS + 128; i++) s->state[i] = 0.0f; /* NR:59 */
  /* arm_lms_norm_init_f32 (NR:62): clears numTaps+blockSize-1 state words,
   * energy = 0, x0 = 0; the coefficient array is NOT cleared. */
  s->energy = 0.0f;
  s->x0 = 0.0f;
}

/* arm_lms_norm_f32 (call site NR:73), published algorithm */
static void lms_norm_f32(orc_lms_t *s, const float *src, const float *ref,
                         float *out, float *err, uint32_t n) {
  float *state = s->state;
  float *cur = &s->state[LMS_TAPS - 1];
  float energy = s->energy, x0 = s->x0, mu = s->mu;
  for (uint32_t b = 0; b < n; b++) {
    float in = src[b];
    *cur++ = in;
    energy -= x0 * x0;
    energy += in * in;
    float sum = 0.0f;
    for (int i = 0; i < LMS_TAPS; i++) sum += state[i] * s->coeffs[i];
    float d = ref[b];
    out[b] = sum; /* may alias src[b]: `in` was read first */
    float e = d - sum;
    err[b] = e;
    float w = (e * mu) / (energy + 0.000000119209289f);
    for (int i = 0; i < LMS_TAPS; i++) s->coeffs[i] += w * state[i];
    x0 = state[0];
    state++;
  }
  s->energy = energy;
  s->x0 = x0;
  /* keep the last numTaps-1 samples at the head of the state buffer */
  memmove(s->state, state, (LMS_TAPS - 1) * sizeof(float));
}

/* the routine alone, for the known answers the reference's compiled arm_lms_norm_f32 gives (tests/test_firmware_kat.py):
 * one call of n samples on an instance given as plain arrays -- coeffs[96] in/out, state[96 - 1 + n] in/out (the first
 * 95 entries carry over), energy_x0[2] in/out */
void orc_lms_norm_f32_kat(float mu, float *coeffs, float *state95, float *energy_x0, const float *src, const float *ref,
                          float *out, float *err, uint32_t n) {
  static orc_lms_t s;
  if (n > (uint32_t)LMS_DELAY) return;
  memset(&s, 0, sizeof(s));
  s.mu = mu;
  s.energy = energy_x0[0];
  s.x0 = energy_x0[1];
  memcpy(s.coeffs, coeffs, LMS_TAPS * sizeof(float));
  memcpy(s.state, state95, (LMS_TAPS - 1) * sizeof(float));
  lms_norm_f32(&s, src, ref, out, err, n);
  memcpy(coeffs, s.coeffs, LMS_TAPS * sizeof(float));
  memcpy(state95, s.state, (LMS_TAPS - 1) * sizeof(float));
  energy_x0[0] = s.energy;
  energy_x0[1] = s.x0;
}

/* NR:66-80 with the ring statics made per-instance */
static void lms_noise_reduction(orc_lms_t *s, int n, float *nrbuffer, float *errsig_out) {
  memcpy(&s->delay[s->inbuf], nrbuffer, (size_t)n * sizeof(float)); /* NR:71 */
  lms_norm_f32(s, nrbuffer, &s->delay[s->outbuf], nrbuffer, s->errsig, (uint32_t)n); /* NR:73 */
  if (errsig_out) memcpy(errsig_out, s->errsig, (size_t)n * sizeof(float));
  s->inbuf += (uint32_t)n;             /* NR:76 */
  s->outbuf = s->inbuf + (uint32_t)n;  /* NR:77 */
  s->inbuf %= 256;                     /* NR:78 */
  s->outbuf %= 256;                    /* NR:79 */
}

/* ------------------------------------------------------------------------ */
/* chain object: the reference's globals (CONV:34-80, NR:18-32, SPEC:109)    */
/* ------------------------------------------------------------------------ */
/* ======================================================================== */
/* F3: biquad cascades.  arm_biquad_cascade_df1_f32 role (the engine's audio filter bank,
 * {b0,b1,b2,a1,a2} x 4 stages in the firmware image); Teensy's fixed-point
 * AudioFilterBiquad (INO:58-59,155-156) follows further down.  Neither library is in the tree.  arm_biquad_cascade_df1_f32 as
 * CMSIS-DSP publishes it: direct form 1 in float, per stage
 *     acc = (b0 * Xn) + (b1 * Xn1) + (b2 * Xn2) + (a1 * Yn1) + (a2 * Yn2)
 * left to right, every product rounded before it is added (this file is built with -ffp-contract=off), feedback
 * terms added.  The routine is in the reference's firmware image as 5 VMUL + 4 VADD per sample, unfused
 * (tests/test_firmware_tables.py).                                                           */
/* ======================================================================== */
void orc_biquad_init(orc_biquad_t *b, int n_stages, const float *coef5) {
  memset(b, 0, sizeof(*b));
  b->n_stages = n_stages < 0 ? 0 : (n_stages > ORC_BIQUAD_MAX_STAGES ? ORC_BIQUAD_MAX_STAGES : n_stages);
  for (int s = 0; s < ORC_BIQUAD_MAX_STAGES; s++) {
    float *c = b->coef + 5 * s;
    if (s < b->n_stages && coef5) memcpy(c, coef5 + 5 * s, 5 * sizeof(float));
    else { c[0] = 1.0f; c[1] = c[2] = c[3] = c[4] = 0.0f; } /* pass-through */
  }
}
void orc_biquad_set_stage(orc_biquad_t *b, int stage, const float *coef5) {
  if (stage < 0 || stage >= ORC_BIQUAD_MAX_STAGES) return;
  memcpy(b->coef + 5 * stage, coef5, 5 * sizeof(float));
  if (stage >= b->n_stages) b->n_stages = stage + 1;
}
void orc_biquad_run(orc_biquad_t *b, float *x, int n) {
  for (int i = 0; i < n; i++) {
    float v = x[i];
    for (int s = 0; s < ORC_BIQUAD_MAX_STAGES; s++) {
      const float *c = b->coef + 5 * s;
      float *st = b->state + 4 * s; /* x1, x2, y1, y2 */
      float y = (c[0] * v) + (c[1] * st[0]) + (c[2] * st[1]) + (c[3] * st[2]) + (c[4] * st[3]);
      st[1] = st[0]; st[0] = v;
      st[3] = st[2]; st[2] = y;
      v = y;
    }
    x[i] = v;
  }
}
/* RBJ audio-EQ sections for the float cascade (a design helper of this build, e.g. for single sections of
 * the engine's bank; AudioFilterBiquad's own setters are orc_teensy_biquad_design below): double design narrowed
 * to float, feedback coefficients stored negated like CMSIS.  kind: 0 LP, 1 HP, 2 BP, 3 notch. */
void orc_biquad_design(int kind, double freq, double q, double fs, float *coef5) {
  const double w0 = freq * (2.0 * 3.14159265358979323846 / fs);
  const double sinW0 = sin(w0), alpha = sinW0 / (q * 2.0), cosW0 = cos(w0);
  const double scale = 1.0 / (1.0 + alpha);
  double b0, b1, b2;
  switch (kind) {
    case 0: b0 = ((1.0 - cosW0) / 2.0) * scale; b1 = (1.0 - cosW0) * scale; b2 = b0; break;
    case 1: b0 = ((1.0 + cosW0) / 2.0) * scale; b1 = -(1.0 + cosW0) * scale; b2 = b0; break;
    case 2: b0 = alpha * scale; b1 = 0.0; b2 = -alpha * scale; break;
    default: b0 = scale; b1 = (-2.0 * cosW0) * scale; b2 = b0; break;
  }
  coef5[0] = (float)b0; coef5[1] = (float)b1; coef5[2] = (float)b2;
  coef5[3] = (float)(-((-2.0 * cosW0) * scale));
  coef5[4] = (float)(-((1.0 - alpha) * scale));
}
/* ---- AudioFilterBiquad of the Teensy Audio library (INO:58-59,75-78,155-156), as the library publishes it --------
 * filter_biquad.{h,cpp}: a FIXED-POINT cascade.  Coefficients are int32 scaled by 2^30 (a1, a2 stored negated); the
 * five products of a sample are 32 x 16 multiplies that keep the top 32 of 48 bits (SMLAWB / SMLAWT:
 * signed_multiply_accumulate_32x16b / t), accumulated on top of the 14 fractional bits the previous sample left
 * behind (`sum &= 0x3FFF`: first-order error feedback), and the output is `signed_saturate_rshift(sum, 16, 14)`.
 * update() works on pairs of samples packed in 32-bit words; unpacked, that is the per-sample recursion below (the
 * second sample of a pair takes the first one's saturated output as y[n-1], exactly as `a1 * out2` does there).
 * The reference's firmware image holds the routine: UBFX, SMLAWB, SMLAWT, SMLAWB, SMLAWT, SMLAWB, SSAT #16 ASR #14,
 * UBFX, SMLAWT, SMLAWB, SMLAWT, SMLAWB, SMLAWT, SSAT #16 ASR #14, PKHBT, UBFX (tests/test_firmware_tables.py), and run
 * under an interpreter it gives this restatement's outputs bit for bit (tests/test_firmware_kat.py).  A stage that was never set has all-zero coefficients
 * ("by default, the filter will not pass anything"); update() always runs stage 0 and goes on to stage s + 1 only
 * if setCoefficients(s + 1) was ever called (it sets the hand-on bit in stage s). */
void orc_teensy_biquad_init(orc_teensy_biquad_t *b) { memset(b, 0, sizeof(*b)); }
/* setCoefficients(uint32_t stage, const int *coefficients) */
void orc_teensy_biquad_setCoefficients_int(orc_teensy_biquad_t *b, int stage, const int32_t *c) {
  if (stage < 0 || stage >= 4) return;
  b->coef[stage][0] = c[0];
  b->coef[stage][1] = c[1];
  b->coef[stage][2] = c[2];
  b->coef[stage][3] = (int32_t)(0u - (uint32_t)c[3]); /* `*coefficients++ * -1` */
  b->coef[stage][4] = (int32_t)(0u - (uint32_t)c[4]);
  b->sum[stage] = 0;                                 /* `*dest &= 0x80000000`: the residue goes, x and y history stay */
  if (stage > 0) b->chained[stage - 1] = 1;          /* `if (stage > 0) *(dest - 1) |= 0x80000000` */
}
/* setCoefficients(uint32_t stage, const double *coefficients): each times 1073741824.0, converted to int */
void orc_teensy_biquad_setCoefficients(orc_teensy_biquad_t *b, int stage, const double *c) {
  int32_t ci[5];
  for (int i = 0; i < 5; i++) ci[i] = (int32_t)(c[i] * 1073741824.0);
  orc_teensy_biquad_setCoefficients_int(b, stage, ci);
}
/* setLowpass / setHighpass / setBandpass / setNotch (filter_biquad.h; the RBJ cookbook in double, the angle
 * `frequency * (2 * 3.141592654 / AUDIO_SAMPLE_RATE_EXACT)` in double too: the five integers the reference's setup() holds
 * for setHighpass(0, 500, 0.5), folded at compile time, come out of exactly this).  kind 0 LP, 1 HP, 2 BP, 3 notch */
void orc_teensy_biquad_design(int kind, float frequency, float q, float fs, int32_t *coef5) {
  const double w0 = (double)frequency * (2.0 * 3.141592654 / (double)fs);
  const double sinW0 = sin(w0);
  const double alpha = sinW0 / ((double)q * 2.0);
  const double cosW0 = cos(w0);
  const double scale = 1073741824.0 / (1.0 + alpha);
  switch (kind) {
    case 0: coef5[0] = (int32_t)(((1.0 - cosW0) / 2.0) * scale); coef5[1] = (int32_t)((1.0 - cosW0) * scale); coef5[2] = coef5[0]; break;
    case 1: coef5[0] = (int32_t)(((1.0 + cosW0) / 2.0) * scale); coef5[1] = (int32_t)(-(1.0 + cosW0) * scale); coef5[2] = coef5[0]; break;
    case 2: coef5[0] = (int32_t)(alpha * scale); coef5[1] = 0; coef5[2] = (int32_t)((-alpha) * scale); break;
    default: coef5[0] = (int32_t)scale; coef5[1] = (int32_t)((-2.0 * cosW0) * scale); coef5[2] = coef5[0]; break;
  }
  coef5[3] = (int32_t)((-2.0 * cosW0) * scale);
  coef5[4] = (int32_t)((1.0 - alpha) * scale);
}
static inline int32_t orc_smlaw(int32_t acc, int32_t c, int16_t v) { /* acc + ((c * v) >> 16), 32-bit wrap-around */
  return (int32_t)((uint32_t)acc + (uint32_t)(int32_t)(((int64_t)c * (int64_t)v) >> 16));
}
/* one update(): n samples of one channel in place, stage after stage */
void orc_teensy_biquad_update(orc_teensy_biquad_t *b, int16_t *data, int n) {
  for (int st = 0; st < 4; st++) { /* do { ... flag = *state & 0x80000000; ... } while (flag); */
    if (st > 0 && !b->chained[st - 1]) break;
    const int32_t *c = b->coef[st];
    int16_t x1 = b->x1[st], x2 = b->x2[st], y1 = b->y1[st], y2 = b->y2[st];
    int32_t sum = b->sum[st] & 0x3FFF;
    for (int i = 0; i < n; i++) {
      const int16_t x = data[i];
      sum = orc_smlaw(sum, c[0], x);
      sum = orc_smlaw(sum, c[1], x1);
      sum = orc_smlaw(sum, c[2], x2);
      sum = orc_smlaw(sum, c[3], y1);
      sum = orc_smlaw(sum, c[4], y2);
      int32_t y = sum >> 14;                       /* signed_saturate_rshift(sum, 16, 14) */
      y = y > 32767 ? 32767 : (y < -32768 ? -32768 : y);
      sum &= 0x3FFF;
      x2 = x1; x1 = x;
      y2 = y1; y1 = (int16_t)y;
      data[i] = (int16_t)y;
    }
    b->x1[st] = x1; b->x2[st] = x2; b->y1[st] = y1; b->y2[st] = y2; b->sum[st] = sum;
  }
}

/* 8th-order Butterworth band-pass f1..f2 as four biquads (the shape SURVEY Appendix C reads out
 * of the firmware image: -3 dB at 150 Hz and at 2.1 ... 3.9 kHz): analog 4th-order prototype,
 * LP -> BP, bilinear transform with pre-warped edges; each section has one zero at z = 1 and one
 * at z = -1; the cascade is normalised to unit gain at sqrt(f1 f2) with the gain spread evenly. */
void orc_design_butter_bp8(double f1, double f2, double fs, float *coef20) {
  const double pi = 3.14159265358979323846;
  const double w1 = 2.0 * fs * tan(pi * f1 / fs), w2 = 2.0 * fs * tan(pi * f2 / fs);
  const double bw = w2 - w1, w0sq = w1 * w2;
  double pr[4], pi_[4]; /* one pole of each conjugate pair in the z plane */
  int np = 0;
  for (int k = 0; k < 2; k++) { /* prototype poles in the upper half plane: k = 0, 1 */
    const double th = pi * (2.0 * k + 1.0 + 4.0) / 8.0;
    const double ar = cos(th) * bw * 0.5, ai = sin(th) * bw * 0.5; /* p * bw / 2 */
    /* s = a +- sqrt(a^2 - w0^2) */
    const double dr = ar * ar - ai * ai - w0sq, di = 2.0 * ar * ai;
    const double mag = sqrt(sqrt(dr * dr + di * di)), ang = 0.5 * atan2(di, dr);
    const double sr = mag * cos(ang), si = mag * sin(ang);
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      double xr = ar + sgn * sr, xi = ai + sgn * si; /* analog pole */
      if (xi < 0) xi = -xi;                            /* keep the upper-half representative */
      /* z = (2 fs + s) / (2 fs - s) */
      const double nr = 2.0 * fs + xr, ni = xi, dr2 = 2.0 * fs - xr, di2 = -xi;
      const double den = dr2 * dr2 + di2 * di2;
      pr[np] = (nr * dr2 + ni * di2) / den;
      pi_[np] = (ni * dr2 - nr * di2) / den;
      np++;
    }
  }
  /* gain of the un-normalised cascade at the centre frequency */
  const double wc = 2.0 * pi * sqrt(f1 * f2) / fs;
  double gr = 1.0, gi = 0.0;
  for (int s = 0; s < 4; s++) {
    const double a1 = -2.0 * pr[s], a2 = pr[s] * pr[s] + pi_[s] * pi_[s];
    /* H = (1 - z^-2) / (1 + a1 z^-1 + a2 z^-2) at z = e^{j wc} */
    const double c1 = cos(wc), s1 = -sin(wc), c2 = cos(2 * wc), s2 = -sin(2 * wc);
    const double nr = 1.0 - c2, ni = -s2, dr = 1.0 + a1 * c1 + a2 * c2, di = a1 * s1 + a2 * s2;
    const double den = dr * dr + di * di;
    const double hr = (nr * dr + ni * di) / den, hi = (ni * dr - nr * di) / den;
    const double tr = gr * hr - gi * hi, ti = gr * hi + gi * hr;
    gr = tr; gi = ti;
  }
  const double g = pow(1.0 / sqrt(gr * gr + gi * gi), 0.25);
  for (int s = 0; s < 4; s++) {
    coef20[5 * s + 0] = (float)g;
    coef20[5 * s + 1] = 0.0f;
    coef20[5 * s + 2] = (float)(-g);
    coef20[5 * s + 3] = (float)(2.0 * pr[s]);                             /* -a1 */
    coef20[5 * s + 4] = (float)(-(pr[s] * pr[s] + pi_[s] * pi_[s]));      /* -a2 */
  }
}

struct orc_chain {
  orc_config_t cfg;
  uint32_t fft_l, hop; /* FFT_length, BUFFER_SIZE*N_BLOCKS */
  double fs_out;       /* SAMPLE_RATE of the CONV stage */
  /* front end (build-defined) */
  uint32_t dphi;
  uint64_t n_in; /* absolute input sample counter */
  float *fir_taps;
  float *fir_ring_re, *fir_ring_im; /* circular, length fir_taps */
  uint32_t fir_pos;
  /* CONV globals */
  float *FFT_buffer, *iFFT_buffer, *FIR_filter_mask;
  float *float_buffer_L, *float_buffer_R, *last_L, *last_R;
  float *mag; /* FFTBufferMag, SPEC:55 */
  double *coef_I, *coef_Q;
  uint8_t first_block;
  uint32_t fill; /* samples gathered in float_buffer_* */
  int oldNRLevel; /* CONV:80 */
  float NFloor;   /* SPEC:109 */
  orc_lms_t nr, als;
  float agc_g;
  float am_dc;
  /* F3 (build-defined): pre-processor IQ swap, noise blanker */
  int swap_iq;
  int iq_slip;            /* +1: the I rail is taken one sample late, -1: the Q rail (I2S channel slip) */
  int16_t slip_i, slip_q; /* the previous raw sample */
  int literal_nr_first_block; /* CONV:326-337 as written for N_BLOCKS > 1: NR on the first 128 samples of a hop only */
  int literal_resynthesis; /* SPEC:221-235 as written: atan2 + table-interpolated arm_cos_f32 / arm_sin_f32 */
  int literal_filter_off;  /* CONV:303 as written: only FFT_length FLOATS (half the spectrum) are copied */
  int nb_on;
  float nb_thr;   /* threshold as a power ratio, 10^(dB/10) */
  float nb_level; /* reference power: smoothed mean |x|^2 of the past windows */
  float nb_acc;   /* post-blanking power accumulated in the current window */
  uint32_t nb_fill;
  /* F3: SAM demodulator (PLL) */
  float sam_g1, sam_g2, sam_wmin, sam_wmax;
  float sam_phs, sam_omega, sam_fil, sam_dc;
  /* F3: the engine's IIR audio filter bank (build-defined) */
  int iir_on;
  orc_biquad_t iir;
};

uint32_t orc_demod_tuning_offset(int demod) {
  /* what AudioSDR::setDemodMode returns (INO:139): the engine is not in the tree but in the firmware image, and run
   * there it answers IF centre 6890 Hz +- half the band (SSB 3000 Hz, CW 1000 Hz), AM / SAM at the centre
   * (tests/golden/firmware_kat.npz `engine_tuning_offset`, tests/test_firmware_kat.py) */
  switch (demod) {
    case ORC_DEMOD_LSB: return 8390u;
    case ORC_DEMOD_USB: return 5390u;
    case ORC_DEMOD_CW_LSB: return 7390u;
    case ORC_DEMOD_CW_USB: return 6390u;
    case ORC_DEMOD_AM:
    case ORC_DEMOD_SAM: return 6890u;
    default: return 0u;
  }
}

static void agc_params(int mode, float *attack, float *decay) {
  *attack = 0.6f;
  switch (mode) {
    case ORC_AGC_FAST: *decay = 0.10f; break;
    case ORC_AGC_MEDIUM: *decay = 0.03f; break;
    case ORC_AGC_SLOW: *decay = 0.008f; break;
    default: *decay = 0.0f; break;
  }
}

void orc_Init_LMS_NR(orc_chain_t *c, int strength) { lms_init(&c->nr, strength); }
void orc_Init_ALS(orc_chain_t *c, int strength) { lms_init(&c->als, strength); }
void orc_LMS_NoiseReduction(orc_chain_t *c, int16_t n, float *nrbuffer) {
  lms_noise_reduction(&c->nr, n, nrbuffer, NULL);
}
void orc_set_nr_level(orc_chain_t *c, int lms_nr) { c->cfg.lms_nr = lms_nr; }

/* CONV:187-207: the mask is built from whatever the tap arrays hold (all zero
 * at boot, INO:180 runs before INO:183). */
void orc_doConvolutionalInitialize(orc_chain_t *c) {
  orc_init_filter_mask(c->FIR_filter_mask, c->coef_I, c->coef_Q, c->fft_l);
}

/* CONV:209-224 */
void orc_reInitializeFilter(orc_chain_t *c, double lo, double hi) {
  orc_calc_cplx_FIR_coeffs(c->coef_I, c->coef_Q, (int)(c->fft_l / 2 + 1), lo, hi,
                           c->fs_out, c->cfg.window);
  orc_init_filter_mask(c->FIR_filter_mask, c->coef_I, c->coef_Q, c->fft_l);
}

/* ---- F3: pre-processor swapIQ (INO:118) and the engine's noise blanker (BK_INO:1259-1260,
 * INO:131).  Both live in the AudioSDR library, so the arithmetic is build-defined:
 * the blanker works on the wide-band IQ stream before the mixer, in windows of
 * 256*decim input samples (the chunk the stage works in).  A sample whose power
 * exceeds  level * 10^(dB/10)  is replaced by zero; `level` is the smoothed
 * (alpha 0.2) mean post-blanking power of the previous windows, so every decision
 * inside a window uses state from before the window. */
void orc_set_swap_iq(orc_chain_t *c, int on) { c->swap_iq = on ? 1 : 0; }
/* preProcessor.startAutoI2SerrorDetection() (INO:117) guards against the I2S fault that leaves one
 * rail of the codec stream a sample behind the other; a recording made through such a front end
 * carries it.  Build-defined correction: slip +1 pairs I[n-1] with Q[n] (delays the I rail by one
 * sample), slip -1 pairs I[n] with Q[n-1]; applied to the raw words, before swapIQ and the gains. */
void orc_set_iq_slip(orc_chain_t *c, int slip) { c->iq_slip = slip > 0 ? 1 : (slip < 0 ? -1 : 0); }
void orc_set_noise_blanker(orc_chain_t *c, int on, float threshold_db) {
  c->nb_on = on ? 1 : 0;
  c->nb_thr = (float)pow(10.0, (double)threshold_db / 10.0);
}
float orc_chain_nb_level(const orc_chain_t *c) { return c->nb_level; }
/* SDR.setInputGain / setIQgainBalance / setOutputGain / setMute (INO:133-135,177) between calls:
 * the input-side values apply to samples as they arrive (what is already in the decimator's delay
 * line keeps the gains it came in with), the output-side ones from the next output sample on */
/* SDR.setAGCmode / disableAGC (INO:120-121, CTL:196-232) and the spectral stage's switch and level
 * (SPEC:112 iNRLevel; CTL:237-297) between calls: plain configuration, the states (gain, NFloor) stay */
void orc_set_agc_mode(orc_chain_t *c, int mode) { c->cfg.agc_mode = mode; }
/* bFilterEnabled of doConvolutionalProcessing (CONV:228,300): read by every frame */
void orc_set_filter_on(orc_chain_t *c, int on) { c->cfg.filter_on = on ? 1 : 0; }
/* SDR.enableALSfilter / disableALSfilter / setALSfilterNotch / setALSfilterPeak (CTL:250-261): which
 * of the instance's two outputs is taken, or none; the instance's weights and delay line stay */
void orc_set_als_mode(orc_chain_t *c, int mode) { c->cfg.als_mode = mode; }
void orc_set_spectral_nr(orc_chain_t *c, int on, float level) {
  c->cfg.spectral_nr = on;
  c->cfg.spectral_level = level;
}
/* SPEC:221-235 re-synthesises every bin as mag' * (arm_cos_f32(phi) + j arm_sin_f32(phi)), phi = atan2(im, re).
 * In exact arithmetic that is X * mag'/mag, which is what this restatement (and the kernels) evaluate by
 * default.  on = 1 evaluates it AS WRITTEN, with arm_sin_f32 / arm_cos_f32 restated from their published
 * algorithm (CMSIS-DSP FastMathFunctions: 512-entry table of sin(2 pi k / 512), 513 entries, linear
 * interpolation on the fractional index; cos = the same with the argument advanced by a quarter turn), so that
 * the distance between the two forms -- the table's interpolation error, up to (2 pi / 512)^2 / 8 = 1.9e-5 of a
 * bin's magnitude -- can be measured (tests/test_oracle_kat.py).  Test infrastructure only. */
void orc_set_literal_resynthesis(orc_chain_t *c, int on) { c->literal_resynthesis = on ? 1 : 0; }
/* CONV:303 as written: with bFilterEnabled == false the sketch copies FFT_length floats, i.e. bins 0 .. FFT_L/2 - 1
 * of the interleaved spectrum; the upper half of iFFT_buffer keeps what the previous frame's in-place inverse
 * transform left there (time-domain samples).  The sketch never takes that branch (INO:198 passes `true`); the
 * restatement (and the product) treat "filter off" as a full bypass.  on = 1 evaluates the line as written, to show
 * what that choice replaces (tests/test_oracle_kat.py).  Test infrastructure only. */
void orc_set_literal_filter_off(orc_chain_t *c, int on) { c->literal_filter_off = on ? 1 : 0; }
/* CONV:326-337 as written: `LMS_NoiseReduction(128, float_buffer_L)` and the `x 1.1; R = L` loop over BUFFER_SIZE touch
 * the first 128 samples of the hop whatever FFT_L is ("apply the LMS but with single block").  For FFT_L = 256
 * (N_BLOCKS = 1, the shipped value) that is the whole hop.  For FFT_L = 512 ... 4096 the other N_BLOCKS - 1 blocks of
 * every hop leave as the filter produced them (L = Re y, R = Im y, no NR, no gain), and the NLMS sees one block in
 * N_BLOCKS, so its 128-sample decorrelation delay becomes a whole hop.  The restatement (and the product) run every
 * block through the stage; on = 1 evaluates the lines as written, to show what that choice replaces
 * (tests/test_oracle_kat.py).  Test infrastructure only. */
void orc_set_literal_nr_first_block(orc_chain_t *c, int on) { c->literal_nr_first_block = on ? 1 : 0; }
static float orc_sin_table[513];
static int orc_sin_table_ready = 0;
static float orc_fast_sin_turns(float in) { /* in: the angle in turns (x / 2 pi), any sign */
  if (!orc_sin_table_ready) {
    /* sinTable_f32 as published: decimal literals with eight places ("0.01227154f"), i.e. the float nearest to the
     * rounded decimal, not to the sine -- the last bit differs for about one entry in five */
    for (int k = 0; k <= 512; k++) {
      char lit[32];
      snprintf(lit, sizeof lit, "%.8f", sin(2.0 * 3.14159265358979323846 * (double)k / 512.0));
      orc_sin_table[k] = strtof(lit, NULL);
    }
    orc_sin_table_ready = 1;
  }
  int32_t n = (int32_t)in;
  if (in < 0.0f) n--;
  in = in - (float)n; /* fractional part, [0, 1) */
  float findex = 512.0f * in;
  uint16_t index = (uint16_t)findex;
  if (index >= 512) { index = 0; findex -= 512.0f; }
  const float fract = findex - (float)index;
  const float a = orc_sin_table[index], b = orc_sin_table[index + 1];
  return (1.0f - fract) * a + fract * b;
}
void orc_arm_sin_table(float *tab513) {
  (void)orc_fast_sin_turns(0.0f);
  memcpy(tab513, orc_sin_table, sizeof(orc_sin_table));
}
float orc_arm_sin_f32(float x) { return orc_fast_sin_turns(x * 0.159154943092f); }
float orc_arm_cos_f32(float x) { return orc_fast_sin_turns(x * 0.159154943092f + 0.25f); }
void orc_set_gains(orc_chain_t *c, float input_gain, float iq_balance, float output_gain, int mute) {
  c->cfg.input_gain = input_gain;
  c->cfg.iq_balance = iq_balance;
  c->cfg.output_gain = output_gain;
  c->cfg.mute = mute ? 1 : 0;
}

/* SAMmode (CTL:384-391) is an AudioSDR demodulator; build-defined here as the
 * classic second-order PLL synchronous detector on the filtered base band y:
 *   corr0 = Re(y e^{-j phs}), corr1 = Im(y e^{-j phs}), det = atan2(corr1, corr0)
 *   weighted by |y|^2/(|y|^2 + 1e-6), omega += g2 det (clamped to +-2 kHz),
 *   phs += previous (g1 det + omega),
 *   audio = corr0 - dc,  dc += (corr0 - dc)/512.
 * Loop constants for zeta = 0.65, omegaN = 200 rad/s at the decimated rate. */
void orc_sam_constants(double fs_out, float *g1, float *g2, float *wmin, float *wmax) {
  const double zeta = 0.65, omegaN = 200.0, fmax = 2000.0;
  const double a = 1.0 - exp(-2.0 * omegaN * zeta / fs_out);
  const double b = -a + 2.0 * (1.0 - exp(-omegaN * zeta / fs_out) * cos(omegaN / fs_out * sqrt(1.0 - zeta * zeta)));
  *g1 = (float)a;
  *g2 = (float)b;
  *wmax = (float)(ORC_TWO_PI * fmax / fs_out);
  *wmin = -*wmax;
}
static void sam_block(orc_chain_t *c, float *L, float *R) {
  const float two_pi = (float)ORC_TWO_PI;
  for (int i = 0; i < ORC_BLOCK; i++) {
    const float sn = sinf(c->sam_phs), cs = cosf(c->sam_phs);
    const float corr0 = L[i] * cs + R[i] * sn;
    const float corr1 = R[i] * cs - L[i] * sn;
    /* the detector fades out below about -60 dBFS (filter start-up, muted input): the
     * angle of rounding residue would otherwise steer the loop at random */
    const float mag2 = corr0 * corr0 + corr1 * corr1;
    const float det = atan2f(corr1, corr0) * (mag2 / (mag2 + 1e-6f));
    const float del_out = c->sam_fil;
    c->sam_omega = c->sam_omega + c->sam_g2 * det;
    if (c->sam_omega < c->sam_wmin) c->sam_omega = c->sam_wmin;
    if (c->sam_omega > c->sam_wmax) c->sam_omega = c->sam_wmax;
    c->sam_fil = c->sam_g1 * det + c->sam_omega;
    c->sam_phs = c->sam_phs + del_out;
    if (c->sam_phs >= two_pi) c->sam_phs -= two_pi;
    if (c->sam_phs < 0.0f) c->sam_phs += two_pi;
    c->sam_dc = c->sam_dc + (corr0 - c->sam_dc) * (1.0f / 512.0f);
    L[i] = corr0 - c->sam_dc;
    R[i] = L[i];
  }
}

/* ---- F2: retune / PBT / mode table (the callers of CONV:209) ------------------- */
/* run-time changes of the engine settings a mode switch touches */
void orc_set_demod(orc_chain_t *c, int demod) { c->cfg.demod = demod; }
/* IIR implementation of the audio filter: 8th-order band-pass f1..f2 at the decimated rate;
 * filter state is cleared like arm_biquad_cascade_df1_init_f32 does */
void orc_set_audio_iir(orc_chain_t *c, int on, double f1, double f2) {
  c->iir_on = on;
  if (on) {
    float coef[20];
    orc_design_butter_bp8(f1, f2, c->fs_out, coef);
    orc_biquad_init(&c->iir, 4, coef);
  }
}
const float *orc_chain_iir_coeffs(const orc_chain_t *c) { return c->iir.coef; }
void orc_set_nco_hz(orc_chain_t *c, double hz) {
  c->cfg.nco_hz = hz;
  c->dphi = (uint32_t)(unsigned long long)llround(hz / c->cfg.fs_in * 4294967296.0);
}

/* checkPBT_Increase (CTL:569-588) / checkPBT_Decrease (CTL:590-612): edge 0 is the
 * LOCUT branch (BUTTON_D3), edge 1 the HICUT branch (BUTTON_D6). */
void orc_pbt_step(double *dFLoCut, double *dFHiCut, int edge, int dir) {
  const double MIN_LOW = 0.0, MAX_LOW = 700.0, MIN_HI = 800.0, MAX_HI = 4000.0; /* GEN:79-82 */
  if (dir > 0) {
    if (edge == 0)
      *dFLoCut = (*dFLoCut + 50) <= MAX_LOW ? (*dFLoCut + 50) : *dFLoCut; /* CTL:574 */
    else
      *dFHiCut = (*dFHiCut + 50) <= MAX_HI ? (*dFHiCut + 50) : *dFHiCut; /* CTL:581 */
  } else {
    if (edge == 0) {
      *dFLoCut = (*dFLoCut - 50) > MIN_LOW ? (*dFLoCut - 50) : *dFLoCut; /* CTL:595 */
      if (*dFLoCut < 0.0) *dFLoCut = 0.0;                                /* CTL:596 */
    } else {
      *dFHiCut = (*dFHiCut - 50) > MIN_HI ? (*dFHiCut - 50) : *dFHiCut; /* CTL:604 */
    }
  }
}

/* build-defined pass band of an audio filter name (CTL:153-177: audioCW "500 Hz",
 * audio2100, audio2700, audio3100, audioAM "3.9 kHz"; SURVEY Appendix C: the
 * engine's band-passes start at 150 Hz) on the side the demodulator listens to */
void orc_passband(int filter, int demod, double *lo, double *hi) {
  double a = 150.0, b = 2700.0;
  switch (filter) {
    case 0: a = 450.0; b = 950.0; break; /* audioCW: 500 Hz around the 700 Hz pitch */
    case 1: b = 2100.0; break;
    case 2: b = 2700.0; break;
    case 3: b = 3100.0; break;
    case 4: b = 3900.0; break;
    case 5: a = 1400.0; b = 1600.0; break; /* audioWSPR, CTL:399 */
    default: break;
  }
  if (demod == ORC_DEMOD_LSB || demod == ORC_DEMOD_CW_LSB) { *lo = -b; *hi = -a; }
  else if (demod == ORC_DEMOD_AM || demod == ORC_DEMOD_SAM) { *lo = -b; *hi = b; }
  else { *lo = a; *hi = b; }
}

/* tuningMode() (CTL:330-423): filter and demodulator of menu entry mndx; returns 0
 * for the entries that need engine features outside the build (5 = SAM). */
int orc_tuning_mode(int mndx, double vfoFreq, int *filter, int *demod) {
  switch (mndx) {
    case 0: *filter = 0; *demod = vfoFreq > 10000000 ? ORC_DEMOD_CW_USB : ORC_DEMOD_CW_LSB; return 1; /* CTL:334-342 */
    case 1: *filter = 1; *demod = vfoFreq > 10000000 ? ORC_DEMOD_CW_USB : ORC_DEMOD_CW_LSB; return 1; /* CTL:345-356 */
    case 2: *filter = 2; *demod = ORC_DEMOD_USB; return 1; /* CTL:358-365 */
    case 3: *filter = 2; *demod = ORC_DEMOD_LSB; return 1; /* CTL:367-374 */
    case 4: *filter = 4; *demod = ORC_DEMOD_AM; return 1;  /* CTL:376-383 */
    case 5: *filter = 4; *demod = ORC_DEMOD_SAM; return 1; /* CTL:385-392 */
    case 6: *filter = 1; *demod = ORC_DEMOD_USB; return 1; /* CTL:404-411 "RTTY" */
    default: return 0;
  }
}

orc_chain_t *orc_chain_create(const orc_config_t *cfg) {
  orc_chain_t *c = (orc_chain_t *)calloc(1, sizeof(*c));
  if (!c) return NULL;
  c->cfg = *cfg;
  c->fft_l = (uint32_t)cfg->fft_l;
  c->hop = c->fft_l / 2;
  int decim = cfg->decim < 1 ? 1 : cfg->decim;
  c->cfg.decim = decim;
  c->fs_out = cfg->fs_in / (double)decim;
  uint32_t n = c->fft_l;
  c->FFT_buffer = (float *)calloc(2 * n, sizeof(float));
  c->iFFT_buffer = (float *)calloc(2 * n, sizeof(float));
  c->FIR_filter_mask = (float *)calloc(2 * n, sizeof(float));
  c->mag = (float *)calloc(2 * n, sizeof(float));
  c->float_buffer_L = (float *)calloc(c->hop, sizeof(float));
  c->float_buffer_R = (float *)calloc(c->hop, sizeof(float));
  c->last_L = (float *)calloc(c->hop, sizeof(float));
  c->last_R = (float *)calloc(c->hop, sizeof(float));
  c->coef_I = (double *)calloc(c->hop + 1, sizeof(double));
  c->coef_Q = (double *)calloc(c->hop + 1, sizeof(double));
  c->first_block = 1; /* CONV:42 */
  c->oldNRLevel = 15; /* CONV:80 */
  c->NFloor = 0.0f;   /* SPEC:109 */
  c->agc_g = 1.0f;
  orc_sam_constants(c->fs_out, &c->sam_g1, &c->sam_g2, &c->sam_wmin, &c->sam_wmax);
  c->am_dc = 0.0f;
  /* NCO: 32-bit phase accumulator, closed form of the absolute sample index */
  {
    double turns = cfg->nco_hz / cfg->fs_in;
    long long q = llround(turns * 4294967296.0);
    c->dphi = (uint32_t)(uint64_t)q;
  }
  /* decimator taps: same windowed-sinc law with FLo=-B, FHi=+B (nFs = 0) */
  if (decim > 1) {
    int nt = cfg->fir_taps;
    double *ti = (double *)calloc((size_t)nt, sizeof(double));
    double *tq = (double *)calloc((size_t)nt, sizeof(double));
    orc_calc_cplx_FIR_coeffs(ti, tq, nt, -cfg->fir_cut_hz, cfg->fir_cut_hz, cfg->fs_in,
                             cfg->window);
    c->fir_taps = (float *)calloc((size_t)nt, sizeof(float));
    for (int i = 0; i < nt; i++) c->fir_taps[i] = (float)ti[i];
    free(ti);
    free(tq);
    c->fir_ring_re = (float *)calloc((size_t)nt, sizeof(float));
    c->fir_ring_im = (float *)calloc((size_t)nt, sizeof(float));
  }
  /* boot order of the sketch: Init_LMS_NR(15) INO:172,
   * doConvolutionalInitialize INO:180, reInitializeFilter INO:183 */
  lms_init(&c->nr, 15);
  lms_init(&c->als, cfg->als_strength > 0 ? cfg->als_strength : 15);
  orc_doConvolutionalInitialize(c);
  orc_reInitializeFilter(c, cfg->flo_hz, cfg->fhi_hz);
  return c;
}

void orc_chain_destroy(orc_chain_t *c) {
  if (!c) return;
  free(c->FFT_buffer); free(c->iFFT_buffer); free(c->FIR_filter_mask); free(c->mag);
  free(c->float_buffer_L); free(c->float_buffer_R); free(c->last_L); free(c->last_R);
  free(c->coef_I); free(c->coef_Q); free(c->fir_taps); free(c->fir_ring_re);
  free(c->fir_ring_im);
  free(c);
}

/* One overlap-save frame: CONV:256-318 (+ SPEC:179-238 when spectral_nr). */
static void conv_frame(orc_chain_t *c) {
  const uint32_t N = c->fft_l, H = c->hop;
  float *F = c->FFT_buffer, *G = c->iFFT_buffer;
  if (c->first_block) { /* CONV:256-263: history = zeros */
    for (uint32_t i = 0; i < N; i++) F[i] = 0.0f;
    c->first_block = 0;
  } else { /* CONV:267-271 */
    for (uint32_t i = 0; i < H; i++) {
      F[2 * i] = c->last_L[i];
      F[2 * i + 1] = c->last_R[i];
    }
  }
  for (uint32_t i = 0; i < H; i++) { /* CONV:274-278 */
    c->last_L[i] = c->float_buffer_L[i];
    c->last_R[i] = c->float_buffer_R[i];
  }
  for (uint32_t i = 0; i < H; i++) { /* CONV:281-285 */
    F[N + 2 * i] = c->float_buffer_L[i];
    F[N + 2 * i + 1] = c->float_buffer_R[i];
  }
  orc_cfft_f32(F, N, 0); /* CONV:291 */

  if (c->cfg.spectral_nr == 2) {
    /* the older variant, backup/RadioDSP_SDR_RX_Conv.ino:1586-1630 (BK_INO): threshold =
     * sum of bins 60..120 (61 terms) / 60 * 3, no smoothing, no state; active when the sketch
     * sets TH_VALUE = 0.8 (nrndx == 3, BK_INO:1346-1350), spectral_level is not read.  Bins
     * scale with FFT_L like the VAD bins of the newer variant; loop limits restated as j < FFT_L;
     * BK_INO:1631-1634 has the mask multiply commented out -- as for the newer variant the build
     * keeps the CONV:301 multiply after it. */
    float *mag = c->mag;
    orc_cmplx_mag_f32(F, mag, N); /* BK_INO:1589 */
    int lo = (int)(60u * N / 256u), hi = (int)(120u * N / 256u);
    float specVal = 0.0f;
    for (int m = lo; m <= hi; m++) specVal = specVal + mag[m]; /* BK_INO:1594 */
    float TH = specVal / (float)(hi - lo);                     /* BK_INO:1595 */
    TH = TH * 3;                                               /* BK_INO:1596 */
    c->NFloor = TH; /* kept only so that the state read-back shows the threshold in use */
    for (uint32_t j = 0; j < N; j++) {                         /* BK_INO:1601-1609 */
      float m0 = mag[j], m1;
      if (m0 <= TH) m1 = (float)((double)m0 * 0.2);
      else m1 = m0 - TH;
      float sc = (m0 > 0.0f) ? (m1 / m0) : 0.0f;               /* BK_INO:1614-1628 */
      F[2 * j] = F[2 * j] * sc;
      F[2 * j + 1] = F[2 * j + 1] * sc;
    }
  } else if (c->cfg.spectral_nr) {
    /* SPEC:182-238, with the out-of-bounds loop limits restated as j < FFT_L
     *  and VAD bins scaled with FFT_L (SPEC:34-35 are for 256). */
    float *mag = c->mag;
    orc_cmplx_mag_f32(F, mag, N); /* SPEC:182 */
    int lo = (int)(30u * N / 256u), hi = (int)(180u * N / 256u);
    float specVal = 0.0f;
    for (int m = lo; m <= hi; m++) specVal = specVal + mag[m]; /* SPEC:194-197 */
    float TH = specVal / (float)(hi - lo);                     /* SPEC:200 */
    TH = (float)((double)TH * ((double)c->cfg.spectral_level * 1.5)); /* SPEC:202 */
    const float beta = 0.65f;                                  /* SPEC:115 */
    c->NFloor += (TH - c->NFloor) * beta;                      /* SPEC:205 */
    c->NFloor = (c->NFloor > 0) ? c->NFloor : 0;               /* SPEC:206 */
    for (uint32_t j = 0; j < N; j++) {                         /* SPEC:210-218 */
      float m0 = mag[j], m1;
      if (m0 <= c->NFloor) m1 = (float)((double)m0 * 0.2);
      else m1 = m0 - c->NFloor;
      if (c->literal_resynthesis) { /* SPEC:221-235 as written (orc_set_literal_resynthesis) */
        const float phi = (float)atan2((double)F[2 * j + 1], (double)F[2 * j]); /* SPEC:229 */
        F[2 * j] = m1 * orc_arm_cos_f32(phi);                                  /* SPEC:231 */
        F[2 * j + 1] = m1 * orc_arm_sin_f32(phi);                              /* SPEC:232 */
        continue;
      }
      /* SPEC:226-235: mag' * (cos phi + j sin phi) == X * mag'/mag */
      float sc = (m0 > 0.0f) ? (m1 / m0) : 0.0f;
      F[2 * j] = F[2 * j] * sc;
      F[2 * j + 1] = F[2 * j + 1] * sc;
    }
  }
  if (c->cfg.filter_on) /* CONV:300-301 */
    orc_cmplx_mult_cmplx_f32(F, c->FIR_filter_mask, G, N);
  else if (c->literal_filter_off) /* CONV:303 as written: FFT_length floats; the rest of iFFT_buffer is stale */
    memcpy(G, F, N * sizeof(float));
  else /* CONV:303 copies only FFT_length floats (bug, never exercised);
          restated as a full bypass */
    memcpy(G, F, 2 * N * sizeof(float));
  orc_cfft_f32(G, N, 1); /* CONV:309 */
  for (uint32_t i = 0; i < H; i++) { /* CONV:314-318 */
    c->float_buffer_L[i] = G[N + 2 * i];
    c->float_buffer_R[i] = G[N + 2 * i + 1];
  }
}

/* post-filter stages on one 128-sample block (L, R in place); q: the block's offset inside its hop */
static void post_block(orc_chain_t *c, float *L, float *R, uint32_t q) {
  const orc_config_t *cf = &c->cfg;
  /* demodulator selection (build-defined) */
  if (cf->demod == ORC_DEMOD_AM) {
    float a[ORC_BLOCK], s = 0.0f;
    for (int i = 0; i < ORC_BLOCK; i++) {
      a[i] = sqrtf(L[i] * L[i] + R[i] * R[i]);
      s += a[i];
    }
    float m = s / (float)ORC_BLOCK;
    float dc_new = c->am_dc + 0.25f * (m - c->am_dc);
    for (int i = 0; i < ORC_BLOCK; i++) {
      float dc = c->am_dc + (dc_new - c->am_dc) * ((float)(i + 1) / (float)ORC_BLOCK);
      L[i] = a[i] - dc;
      R[i] = L[i];
    }
    c->am_dc = dc_new;
  } else if (cf->demod == ORC_DEMOD_SAM) {
    sam_block(c, L, R);
  } else if (cf->demod != ORC_DEMOD_IQ) {
    for (int i = 0; i < ORC_BLOCK; i++) R[i] = L[i];
  }
  /* the engine's IIR audio filter (SDR.setAudioFilter, CTL:153-177) when that implementation of
   * the audio filter is selected: mono audio through the biquad cascade */
  if (c->iir_on && cf->demod != ORC_DEMOD_IQ) {
    orc_biquad_run(&c->iir, L, ORC_BLOCK);
    for (int i = 0; i < ORC_BLOCK; i++) R[i] = L[i];
  }
  /* CONV:326-337 (applied per 128-block: deviation from the N_BLOCKS>1 bug, unless orc_set_literal_nr_first_block) */
  if (cf->lms_nr > 0 && !(c->literal_nr_first_block && q != 0)) {
    if (cf->lms_nr != c->oldNRLevel) { /* CONV:327-330 */
      lms_init(&c->nr, cf->lms_nr);
      c->oldNRLevel = cf->lms_nr;
    }
    lms_noise_reduction(&c->nr, ORC_BLOCK, L, NULL); /* CONV:332 */
    for (int i = 0; i < ORC_BLOCK; i++) {            /* CONV:333-336 */
      L[i] = (float)((double)L[i] * 1.1);
      R[i] = L[i];
    }
  }
  /* ALS filter: second NLMS instance; notch = error, peak = prediction */
  if (cf->als_mode != ORC_ALS_OFF) {
    float y[ORC_BLOCK], e[ORC_BLOCK];
    memcpy(y, L, sizeof(y));
    lms_noise_reduction(&c->als, ORC_BLOCK, y, e);
    const float *o = (cf->als_mode == ORC_ALS_NOTCH) ? e : y;
    for (int i = 0; i < ORC_BLOCK; i++) {
      L[i] = o[i];
      R[i] = o[i];
    }
  }
  /* AGC (build-defined): block power -> target gain -> one-pole attack/decay,
   * gain ramped linearly across the block */
  if (cf->agc_mode != ORC_AGC_OFF) {
    float attack, decay;
    agc_params(cf->agc_mode, &attack, &decay);
    float p = 0.0f;
    for (int i = 0; i < ORC_BLOCK; i++) p += L[i] * L[i] + R[i] * R[i];
    p = p / (float)(2 * ORC_BLOCK);
    float rms = sqrtf(p);
    float gt = 0.25f / (rms + 1e-6f);
    if (gt > 100.0f) gt = 100.0f;
    float coef = (gt < c->agc_g) ? attack : decay;
    float g_new = c->agc_g + coef * (gt - c->agc_g);
    for (int i = 0; i < ORC_BLOCK; i++) {
      float g = c->agc_g + (g_new - c->agc_g) * ((float)(i + 1) / (float)ORC_BLOCK);
      L[i] = L[i] * g;
      R[i] = R[i] * g;
    }
    c->agc_g = g_new;
  }
  float og = cf->mute ? 0.0f : cf->output_gain;
  for (int i = 0; i < ORC_BLOCK; i++) {
    L[i] = L[i] * og;
    R[i] = R[i] * og;
  }
}

int orc_chain_process(orc_chain_t *c, const int16_t *iq, int n_blocks,
                      int16_t *out_i16, float *out_f32) {
  const orc_config_t *cf = &c->cfg;
  int produced = 0;
  for (int b = 0; b < n_blocks; b++) {
    for (int i = 0; i < ORC_BLOCK; i++) {
      const int16_t *raw = &iq[2 * ((size_t)b * ORC_BLOCK + (size_t)i)];
      int16_t s[2] = {raw[0], raw[1]};
      if (c->iq_slip > 0) s[0] = c->slip_i;      /* orc_set_iq_slip */
      else if (c->iq_slip < 0) s[1] = c->slip_q;
      c->slip_i = raw[0];
      c->slip_q = raw[1];
      /* arm_q15_to_float, CONV:241-242 */
      float xr = (float)s[c->swap_iq ? 1 : 0] / 32768.0f; /* preProcessor.swapIQ, INO:118 */
      float xi = (float)s[c->swap_iq ? 0 : 1] / 32768.0f;
      xr = xr * cf->iq_balance; /* setIQgainBalance, INO:135 */
      xr = xr * cf->input_gain; /* setInputGain, INO:133 */
      xi = xi * cf->input_gain;
      if (c->nb_on) { /* noise blanker (build-defined, see orc_set_noise_blanker) */
        const float pw = xr * xr + xi * xi;
        if (c->nb_level > 0.0f && pw > c->nb_level * c->nb_thr) {
          xr = 0.0f;
          xi = 0.0f;
          /* what the chain keeps of a sample is the blanked word: a slip correction switched on at the
           * next call pairs its first sample with that (while the correction runs it works on raw words) */
          if (c->iq_slip == 0) c->slip_i = c->slip_q = 0;
        } else {
          c->nb_acc += pw;
        }
        if (++c->nb_fill == 256u * (uint32_t)(cf->decim > 1 ? cf->decim : 1)) {
          const float mean = c->nb_acc / (float)c->nb_fill;
          c->nb_level = (c->nb_level > 0.0f) ? c->nb_level + 0.2f * (mean - c->nb_level) : mean;
          c->nb_acc = 0.0f;
          c->nb_fill = 0;
        }
      }
      uint64_t n = c->n_in++;
      if (c->dphi != 0u) { /* y = x * exp(-j*theta_n) */
        uint32_t ph = (uint32_t)n * c->dphi;
        double th = ORC_TWO_PI * (double)ph / 4294967296.0;
        float co = (float)cos(th), si = (float)sin(th);
        float yr = xr * co + xi * si;
        float yi = xi * co - xr * si;
        xr = yr;
        xi = yi;
      }
      if (cf->decim > 1) {
        uint32_t nt = (uint32_t)cf->fir_taps;
        c->fir_ring_re[c->fir_pos] = xr;
        c->fir_ring_im[c->fir_pos] = xi;
        uint32_t newest = c->fir_pos;
        c->fir_pos = (c->fir_pos + 1) % nt;
        if ((n % (uint64_t)cf->decim) != 0) continue;
        /* y[m] = sum_k h[k] x[4m-k] */
        float ar = 0.0f, ai = 0.0f;
        uint32_t p = newest;
        for (uint32_t k = 0; k < nt; k++) {
          ar += c->fir_taps[k] * c->fir_ring_re[p];
          ai += c->fir_taps[k] * c->fir_ring_im[p];
          p = (p == 0) ? nt - 1 : p - 1;
        }
        xr = ar;
        xi = ai;
      }
      c->float_buffer_L[c->fill] = xr;
      c->float_buffer_R[c->fill] = xi;
      c->fill++;
      if (c->fill == c->hop) {
        c->fill = 0;
        conv_frame(c);
        for (uint32_t q = 0; q < c->hop; q += ORC_BLOCK) {
          float *L = &c->float_buffer_L[q], *R = &c->float_buffer_R[q];
          post_block(c, L, R, q);
          for (int j = 0; j < ORC_BLOCK; j++) {
            if (out_f32) {
              out_f32[2 * (produced + j)] = L[j];
              out_f32[2 * (produced + j) + 1] = R[j];
            }
          }
          if (out_i16) { /* arm_float_to_q15, CONV:346-347 */
            int16_t l16[ORC_BLOCK], r16[ORC_BLOCK];
            orc_float_to_q15(L, l16, ORC_BLOCK);
            orc_float_to_q15(R, r16, ORC_BLOCK);
            for (int j = 0; j < ORC_BLOCK; j++) {
              out_i16[2 * (produced + j)] = l16[j];
              out_i16[2 * (produced + j) + 1] = r16[j];
            }
          }
          produced += ORC_BLOCK;
        }
      }
    }
  }
  return produced;
}

const float *orc_chain_mask(const orc_chain_t *c) { return c->FIR_filter_mask; }
const float *orc_chain_fir_taps(const orc_chain_t *c) { return c->fir_taps; }
const float *orc_chain_lms_coeffs(const orc_chain_t *c, int which) {
  return which ? c->als.coeffs : c->nr.coeffs;
}
float orc_chain_nfloor(const orc_chain_t *c) { return c->NFloor; }
float orc_chain_agc_gain(const orc_chain_t *c) { return c->agc_g; }
uint32_t orc_chain_nco_dphi(const orc_chain_t *c) { return c->dphi; }

int orc_multi_process(const orc_config_t *cfg, int n_ch, const int16_t *iq,
                      int n_blocks, int16_t *out_i16, int n_threads) {
  int decim = cfg->decim < 1 ? 1 : cfg->decim;
  size_t in_stride = (size_t)n_blocks * ORC_BLOCK * 2;
  size_t out_stride = (size_t)n_blocks * ORC_BLOCK / (size_t)decim * 2;
  int produced = 0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int ch = 0; ch < n_ch; ch++) {
    orc_chain_t *c = orc_chain_create(cfg);
    int p = orc_chain_process(c, iq + (size_t)ch * in_stride, n_blocks,
                              out_i16 ? out_i16 + (size_t)ch * out_stride : NULL, NULL);
    orc_chain_destroy(c);
    if (ch == 0) produced = p;
  }
  return produced;
}

/* ======================================================================== */
/* F1: IQ panadapter spectrum, analyze_fft256iq.cpp (FFTIQ)                   */
/* ======================================================================== */

/* q15 window tables.  Teensy Audio's windows.c is not in the tree; the tables the sketch links
 * (AudioWindowHanning256 INO:144, AudioWindowHanning1024 INO:147, AudioWindowBlackmanNuttall256
 * FFTIQ.h:56) are in the reference's shipped firmware image and equal, entry for entry,
 *     w[i] = min(32767, round(32768 * window(i / (N - 1))))
 * (tests/golden/firmware_tables.npz, tests/test_firmware_tables.py).  The remaining windows of
 * FFTIQ.h:30-50 follow the same rule from their textbook definitions (not in the image: unpinned).
 * ids: 0 none, 1 Hanning, 2 BlackmanHarris, 3 BlackmanNuttall, 4 Bartlett, 5 Blackman, 6 Flattop,
 * 7 Nuttall, 8 Welch, 9 Hamming, 10 Cosine, 11 Tukey (alpha 0.5) */
static double orc_window_shape(int id, double x) {
  const double t = ORC_TWO_PI * x;
  switch (id) {
    case 1: return 0.5 - 0.5 * cos(t);
    case 2: return 0.35875 - 0.48829 * cos(t) + 0.14128 * cos(2 * t) - 0.01168 * cos(3 * t);
    case 3: return 0.3635819 - 0.4891775 * cos(t) + 0.1365995 * cos(2 * t) - 0.0106411 * cos(3 * t);
    case 4: return x <= 0.5 ? 2.0 * x : 2.0 - 2.0 * x;
    case 5: return 0.42 - 0.5 * cos(t) + 0.08 * cos(2 * t);
    case 6: return 0.21557895 - 0.41663158 * cos(t) + 0.277263158 * cos(2 * t) - 0.083578947 * cos(3 * t) + 0.006947368 * cos(4 * t);
    case 7: return 0.355768 - 0.487396 * cos(t) + 0.144232 * cos(2 * t) - 0.012604 * cos(3 * t);
    case 8: { const double u = 2.0 * x - 1.0; return 1.0 - u * u; }
    case 9: return 0.54 - 0.46 * cos(t);
    case 10: return sin(0.5 * t);
    case 11: return x < 0.25 ? 0.5 - 0.5 * cos(2 * t) : (x > 0.75 ? 0.5 - 0.5 * cos(2 * ORC_TWO_PI * (1.0 - x)) : 1.0);
    default: return 1.0;
  }
}
void orc_window_q15_n(int window_id, int n, int16_t *w) {
  for (int i = 0; i < n; i++) {
    long q = lround(32768.0 * orc_window_shape(window_id, (double)i / (double)(n - 1)));
    w[i] = (int16_t)(q > 32767 ? 32767 : (q < -32768 ? -32768 : q));
  }
}
void orc_window_q15(int window_id, int16_t *w) { orc_window_q15_n(window_id, 256, w); }

/* ---- arm_cfft_radix4_q15 (FFTIQ.cpp:82; init FFTIQ.h:58: 256 points, forward, bit reversal on) ----
 * CMSIS-DSP is not in the tree.  This is the library's published routine restated: arm_radix4_butterfly_q15 in its
 * DSP-extension form (what a Cortex-M7 build runs: two int16 per 32-bit word, __SHADD16 / __QADD16 / __SMUAD ...
 * written out below on the halves), followed by arm_bitreversal_q15.  What the firmware image pins of it is the
 * twiddle table (twiddleCoef_4096_q15: 3072 (cos, sin) pairs, floor(32768 x) clamped to int16 -- generated here by
 * that rule, compared with the image's table in tests/test_firmware_tables.py) and armBitRevTable; the instruction
 * sequence itself is from the published source and carries no reference-held pin.
 * Scaling as published: the first stage takes its inputs >> 2 and halves once more, every middle stage divides by
 * four, the last by two: 1/N in total (1.15 in, 9.7 out for 256 points, 11.5 for 1024). */
typedef struct { int32_t lo, hi; } orc_pk; /* one packed word: lo = real (even index), hi = imaginary */
static inline int32_t orc_ssat16(int32_t v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
static inline orc_pk pk_qadd16(orc_pk a, orc_pk b) { return (orc_pk){orc_ssat16(a.lo + b.lo), orc_ssat16(a.hi + b.hi)}; }
static inline orc_pk pk_qsub16(orc_pk a, orc_pk b) { return (orc_pk){orc_ssat16(a.lo - b.lo), orc_ssat16(a.hi - b.hi)}; }
static inline orc_pk pk_shadd16(orc_pk a, orc_pk b) { return (orc_pk){(a.lo + b.lo) >> 1, (a.hi + b.hi) >> 1}; }
static inline orc_pk pk_shsub16(orc_pk a, orc_pk b) { return (orc_pk){(a.lo - b.lo) >> 1, (a.hi - b.hi) >> 1}; }
/* exchange forms: the top half of the result pairs a.hi with b.lo, the bottom half a.lo with b.hi */
static inline orc_pk pk_qasx(orc_pk a, orc_pk b) { return (orc_pk){orc_ssat16(a.lo - b.hi), orc_ssat16(a.hi + b.lo)}; }
static inline orc_pk pk_qsax(orc_pk a, orc_pk b) { return (orc_pk){orc_ssat16(a.lo + b.hi), orc_ssat16(a.hi - b.lo)}; }
static inline orc_pk pk_shasx(orc_pk a, orc_pk b) { return (orc_pk){(a.lo - b.hi) >> 1, (a.hi + b.lo) >> 1}; }
static inline orc_pk pk_shsax(orc_pk a, orc_pk b) { return (orc_pk){(a.lo + b.hi) >> 1, (a.hi - b.lo) >> 1}; }
/* out1 = __SMUAD(C, R) >> 16; out2 = __SMUSDX(C, R); word = (out2 & 0xFFFF0000) | (out1 & 0xFFFF), C = (cos, sin):
 * R times exp(-j theta), both components >> 16 (32-bit wrap-around arithmetic like the instructions) */
static inline orc_pk pk_twiddle(orc_pk c, orc_pk r) {
  const int32_t out1 = (int32_t)((uint32_t)(c.lo * r.lo) + (uint32_t)(c.hi * r.hi));
  const int32_t out2 = (int32_t)((uint32_t)(c.lo * r.hi) - (uint32_t)(c.hi * r.lo));
  return (orc_pk){(int16_t)(out1 >> 16), (int16_t)(out2 >> 16)};
}
static inline orc_pk pk_load(const int16_t *p) { return (orc_pk){p[0], p[1]}; }
static inline void pk_store(int16_t *p, orc_pk v) { p[0] = (int16_t)v.lo; p[1] = (int16_t)v.hi; }
/* twiddleCoef_4096_q15[2k], [2k+1] */
static inline orc_pk orc_twiddle_4096(uint32_t k) {
  const double a = ORC_TWO_PI * (double)k / 4096.0;
  double c = floor(32768.0 * cos(a)), s = floor(32768.0 * sin(a));
  return (orc_pk){(int32_t)(c > 32767.0 ? 32767.0 : c), (int32_t)(s > 32767.0 ? 32767.0 : s)};
}
void orc_twiddle_q15_4096(int16_t *table6144) {
  for (uint32_t k = 0; k < 3072; k++) pk_store(table6144 + 2 * k, orc_twiddle_4096(k));
}

static void orc_radix4_butterfly_q15(int16_t *src, uint32_t fftLen, uint32_t twidCoefModifier) {
  const orc_pk zero = {0, 0};
  uint32_t n1, n2 = fftLen >> 2, ic = 0;
  /* first stage: every input >> 2 (two __SHADD16 with zero) */
  for (uint32_t j = 0; j < n2; j++) {
    int16_t *p0 = src + 2 * j, *p1 = p0 + 2 * n2, *p2 = p1 + 2 * n2, *p3 = p2 + 2 * n2;
    orc_pk T = pk_shadd16(pk_shadd16(pk_load(p0), zero), zero);
    orc_pk S = pk_shadd16(pk_shadd16(pk_load(p2), zero), zero);
    orc_pk R = pk_qadd16(T, S);
    S = pk_qsub16(T, S);
    T = pk_shadd16(pk_shadd16(pk_load(p1), zero), zero);
    orc_pk U = pk_shadd16(pk_shadd16(pk_load(p3), zero), zero);
    orc_pk V = pk_qadd16(T, U);
    pk_store(p0, pk_shadd16(R, V));
    R = pk_qsub16(R, V);
    pk_store(p1, pk_twiddle(orc_twiddle_4096(2 * ic), R)); /* co2, si2: the k = 2 output goes to the second quarter */
    T = pk_qsub16(T, U);
    R = pk_qasx(S, T);
    S = pk_qsax(S, T);
    pk_store(p2, pk_twiddle(orc_twiddle_4096(ic), S));     /* co1, si1 */
    pk_store(p3, pk_twiddle(orc_twiddle_4096(3 * ic), R)); /* co3, si3 */
    ic += twidCoefModifier;
  }
  twidCoefModifier <<= 2;
  /* middle stages */
  for (uint32_t k = fftLen / 4; k > 4; k >>= 2) {
    n1 = n2;
    n2 >>= 2;
    ic = 0;
    for (uint32_t j = 0; j <= n2 - 1; j++) {
      const orc_pk C1 = orc_twiddle_4096(ic), C2 = orc_twiddle_4096(2 * ic), C3 = orc_twiddle_4096(3 * ic);
      ic += twidCoefModifier;
      for (uint32_t i0 = j; i0 < fftLen; i0 += n1) {
        int16_t *p0 = src + 2 * i0, *p1 = p0 + 2 * n2, *p2 = p1 + 2 * n2, *p3 = p2 + 2 * n2;
        orc_pk T = pk_load(p0), S = pk_load(p2);
        orc_pk R = pk_qadd16(T, S);
        S = pk_qsub16(T, S);
        T = pk_load(p1);
        orc_pk U = pk_load(p3);
        orc_pk V = pk_qadd16(T, U);
        pk_store(p0, pk_shadd16(pk_shadd16(R, V), zero));
        R = pk_shsub16(R, V);
        pk_store(p1, pk_twiddle(C2, R));
        T = pk_qsub16(T, U);
        R = pk_shasx(S, T);
        S = pk_shsax(S, T);
        pk_store(p2, pk_twiddle(C1, S));
        pk_store(p3, pk_twiddle(C3, R));
      }
    }
    twidCoefModifier <<= 2;
  }
  /* last stage: no twiddles */
  for (uint32_t j = 0; j < (fftLen >> 2); j++) {
    int16_t *p = src + 8 * j;
    const orc_pk xa = pk_load(p), xb = pk_load(p + 2), xc = pk_load(p + 4), xd = pk_load(p + 6);
    const orc_pk R = pk_qadd16(xa, xc), T = pk_qadd16(xb, xd);
    pk_store(p, pk_shadd16(R, T));
    pk_store(p + 2, pk_shsub16(R, T));
    const orc_pk S = pk_qsub16(xa, xc), U = pk_qsub16(xb, xd);
    pk_store(p + 4, pk_shsax(S, U));
    pk_store(p + 6, pk_shasx(S, U));
  }
}
/* arm_bitreversal_q15 walks armBitRevTable and swaps pairs; its net effect -- element i <-> element bit-reverse(i)
 * over log2(n) bits -- is written directly (the table-driven walk is replayed with the image's own table in
 * tests/test_firmware_tables.py) */
static void orc_bitreversal_q15(int16_t *src, uint32_t n) {
  uint32_t bits = 0;
  while ((1u << bits) < n) bits++;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < bits; b++) r |= ((i >> b) & 1u) << (bits - 1 - b);
    if (r > i) {
      int16_t t0 = src[2 * i], t1 = src[2 * i + 1];
      src[2 * i] = src[2 * r]; src[2 * i + 1] = src[2 * r + 1];
      src[2 * r] = t0; src[2 * r + 1] = t1;
    }
  }
}
/* n = 256 or 1024 (twidCoefModifier 4096 / n as arm_cfft_radix4_init_q15 sets it), interleaved re/im int16 */
void orc_cfft_radix4_q15_n(int16_t *buf, int n) {
  orc_radix4_butterfly_q15(buf, (uint32_t)n, 4096u / (uint32_t)n);
  orc_bitreversal_q15(buf, (uint32_t)n);
}
void orc_cfft_radix4_q15_256(int16_t *buf) { orc_cfft_radix4_q15_n(buf, 256); }

/* exact floor(sqrt(x)): not what the reference runs, kept to state how far its approximation sits from it */
uint32_t orc_sqrt_uint32(uint32_t x) {
  uint32_t r = (uint32_t)sqrt((double)x);
  while ((uint64_t)r * r > x) r--;
  while ((uint64_t)(r + 1) * (r + 1) <= x) r++;
  return r;
}

/* sqrt_uint32_approx (FFTIQ.cpp:105), Teensy Audio's utility/sqrt_integer.h as published: a first guess from a
 * 33-entry table indexed by the count of leading zeros, then two Newton steps in integer arithmetic.  The table is
 * the one in the reference's firmware image (offset 0x1f558; tests/golden/firmware_tables.npz).  in = 0 reads the
 * guess 0 and divides by it; UDIV by zero yields 0 on the Cortex-M7 (DIV_0_TRP clear), so the result is 0. */
static const uint16_t orc_sqrt_guess[33] = {55109, 38968, 27555, 19484, 13778, 9742, 6889, 4871, 3445, 2436, 1723,
                                            1218,  862,   609,   431,   305,   216,  153,  108,  77,   54,   39,
                                            27,    20,    14,    10,    7,     5,    4,    3,    2,    1,    0};
const uint16_t *orc_sqrt_guess_table(void) { return orc_sqrt_guess; }
static inline uint32_t orc_udiv(uint32_t a, uint32_t b) { return b ? a / b : 0; }
uint32_t orc_sqrt_uint32_approx(uint32_t in) {
  uint32_t n = orc_sqrt_guess[in ? __builtin_clz(in) : 32];
  n = (orc_udiv(in, n) + n) / 2;
  n = (orc_udiv(in, n) + n) / 2;
  return n;
}

struct orc_fft256iq {
  int16_t window[256];
  int has_window;
  int16_t prev_i[128], prev_q[128];
  int have_prev;
  int16_t buffer[512]; /* FFTIQ.h:103 */
  uint32_t sum[256];   /* FFTIQ.h:104 */
  uint8_t count, naverage;
  uint16_t output[256]; /* FFTIQ.h:99 */
};

orc_fft256iq_t *orc_fft256iq_create(int naverage, int window_id) {
  orc_fft256iq_t *s = (orc_fft256iq_t *)calloc(1, sizeof(*s));
  s->naverage = (uint8_t)(naverage <= 0 ? 1 : naverage); /* averageTogether, FFTIQ.h:88-91 */
  s->has_window = window_id != 0;
  orc_window_q15(window_id, s->window);
  return s;
}
void orc_fft256iq_destroy(orc_fft256iq_t *s) { free(s); }
/* averageTogether(uint8_t n), FFTIQ.h:88-91: count is left alone */
void orc_fft256iq_averageTogether(orc_fft256iq_t *s, int n) { s->naverage = (uint8_t)(n <= 0 ? 1 : n); }
/* windowFunction(const int16_t *w), FFTIQ.h:93-95 */
void orc_fft256iq_windowFunction(orc_fft256iq_t *s, int window_id) {
  s->has_window = window_id != 0;
  orc_window_q15(window_id, s->window);
}
/* the reference's own signature: any table, NULL = no window (`if (window)`, FFTIQ.cpp:81) */
void orc_fft256iq_windowFunction_table(orc_fft256iq_t *s, const int16_t *w) {
  s->has_window = w != NULL;
  if (w) memcpy(s->window, w, sizeof(s->window));
}
/* float read(unsigned int binNumber), FFTIQ.h:70-73 */
float orc_fft256iq_read(const orc_fft256iq_t *s, unsigned int binNumber) {
  if (binNumber > 255) return 0.0;
  return (float)(s->output[binNumber]) * (1.0 / 16384.0);
}
/* float read(unsigned int binFirst, unsigned int binLast), FFTIQ.h:75-86, as written */
float orc_fft256iq_read_range(const orc_fft256iq_t *s, unsigned int binFirst, unsigned int binLast) {
  if (binFirst > binLast) {
    unsigned int tmp = binLast;
    binLast = binFirst;
    binFirst = tmp;
  }
  if (binFirst > 255) return 0.0;
  if (binLast > 255) binLast = 255;
  uint32_t sum = 0;
  do {
    sum += s->output[binFirst++];
  } while (binFirst < binLast);
  return (float)sum * (1.0 / 16384.0);
}
const uint16_t *orc_fft256iq_output(const orc_fft256iq_t *s) { return s->output; }

int orc_fft256iq_update(orc_fft256iq_t *s, const int16_t *bi, const int16_t *bq) {
  if (!s->have_prev) { /* FFTIQ.cpp:73-77 */
    memcpy(s->prev_i, bi, sizeof(s->prev_i));
    memcpy(s->prev_q, bq, sizeof(s->prev_q));
    s->have_prev = 1;
    return 0;
  }
  /* copy_to_fft_buffer, FFTIQ.cpp:38-48: word = I | Q << 16 */
  for (int i = 0; i < 128; i++) {
    s->buffer[2 * i] = s->prev_i[i];
    s->buffer[2 * i + 1] = s->prev_q[i];
    s->buffer[256 + 2 * i] = bi[i];
    s->buffer[256 + 2 * i + 1] = bq[i];
  }
  if (s->has_window) { /* apply_window_to_fft_buffer, FFTIQ.cpp:50-63 */
    for (int i = 0; i < 256; i++) {
      s->buffer[2 * i] = (int16_t)(((int32_t)s->buffer[2 * i] * s->window[i]) >> 15);
      s->buffer[2 * i + 1] = (int16_t)(((int32_t)s->buffer[2 * i + 1] * s->window[i]) >> 15);
    }
  }
  orc_cfft_radix4_q15_256(s->buffer); /* FFTIQ.cpp:82 */
  for (int i = 0; i < 256; i++) { /* FFTIQ.cpp:86-98 */
    int32_t r = s->buffer[2 * i], q = s->buffer[2 * i + 1];
    uint32_t magsq = (uint32_t)(r * r + q * q); /* multiply_16tx16t_add_16bx16b */
    if (s->count == 0) s->sum[i] = magsq / s->naverage;
    else s->sum[i] += magsq / s->naverage;
  }
  int fresh = 0;
  if (++s->count == s->naverage) { /* FFTIQ.cpp:99-113 */
    s->count = 0;
    for (int i = 0; i < 256; i++) s->output[255 - (i ^ 128)] = (uint16_t)orc_sqrt_uint32_approx(s->sum[i]);
    fresh = 1;
  }
  memcpy(s->prev_i, bi, sizeof(s->prev_i)); /* FFTIQ.cpp:114-117 */
  memcpy(s->prev_q, bq, sizeof(s->prev_q));
  return fresh;
}

/* ======================================================================== */
/* AudioAnalyzeFFT1024 (Teensy Audio library; `AudioAnalyzeFFT1024 AudioFFT` fed from Q_out_L,    */
/* INO:57,87, read by the display).  Not in the tree: restated from the library's published      */
/* update(): blocks are collected eight at a time with four kept (1024-sample frames, hop 512),  */
/* real samples with zero imaginary parts, q15 window (x*w)>>15, arm_cfft_radix4_q15 (here the   */
/* same arm_cfft_radix4_q15 restatement as F1, five stages), output[i] = sqrt_uint32_approx(re^2 */
/* + im^2) for the 512 bins at and above DC; no averaging.                                       */
/* ======================================================================== */
struct orc_fft1024 {
  int16_t window[1024];
  int has_window;
  int16_t blocks[8][128]; /* blocklist[8] */
  int state;
  uint16_t output[512];
};
orc_fft1024_t *orc_fft1024_create(int window_id) {
  orc_fft1024_t *s = (orc_fft1024_t *)calloc(1, sizeof(*s));
  s->has_window = window_id != 0;
  orc_window_q15_n(window_id, 1024, s->window);
  return s;
}
void orc_fft1024_destroy(orc_fft1024_t *s) { free(s); }
void orc_fft1024_windowFunction_table(orc_fft1024_t *s, const int16_t *w) {
  s->has_window = w != NULL;
  if (w) memcpy(s->window, w, sizeof(s->window));
}
const uint16_t *orc_fft1024_output(const orc_fft1024_t *s) { return s->output; }
/* one update() tick with one 128-sample block; 1 when output[] is fresh */
int orc_fft1024_update(orc_fft1024_t *s, const int16_t *block) {
  memcpy(s->blocks[s->state], block, 128 * sizeof(int16_t));
  if (s->state < 7) { s->state++; return 0; }
  int16_t buf[2048];
  for (int b = 0; b < 8; b++)
    for (int i = 0; i < 128; i++) {
      int32_t v = s->blocks[b][i];
      if (s->has_window) v = (v * s->window[b * 128 + i]) >> 15;
      buf[2 * (b * 128 + i)] = (int16_t)v;
      buf[2 * (b * 128 + i) + 1] = 0;
    }
  orc_cfft_radix4_q15_n(buf, 1024);
  for (int i = 0; i < 512; i++) {
    int32_t r = buf[2 * i], q = buf[2 * i + 1];
    s->output[i] = (uint16_t)orc_sqrt_uint32_approx((uint32_t)(r * r + q * q));
  }
  for (int b = 0; b < 4; b++) memcpy(s->blocks[b], s->blocks[b + 4], 128 * sizeof(int16_t));
  s->state = 4;
  return 1;
}

/* many channels of the F1 analyser (cpu_baseline leg); returns a checksum of the
 * last spectrum of every channel so the work cannot be optimised away */
uint64_t orc_fft256iq_multi(int naverage, int window_id, int n_ch, const int16_t *iq, int n_blocks,
                            int n_threads) {
  uint64_t total = 0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
  for (int c = 0; c < n_ch; c++) {
    orc_fft256iq_t *s = orc_fft256iq_create(naverage, window_id);
    const int16_t *row = iq + (size_t)c * n_blocks * 128 * 2;
    int16_t bi[128], bq[128];
    for (int b = 0; b < n_blocks; b++) {
      for (int i = 0; i < 128; i++) {
        bi[i] = row[2 * (b * 128 + i)];
        bq[i] = row[2 * (b * 128 + i) + 1];
      }
      orc_fft256iq_update(s, bi, bq);
    }
    for (int i = 0; i < 256; i++) total += s->output[i];
    orc_fft256iq_destroy(s);
  }
  return total;
}
