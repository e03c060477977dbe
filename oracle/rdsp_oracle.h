/*
 * rdsp_oracle.h -- CPU ORACLE for the per-block IQ receive chain.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke()
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py may load
 * it.  The product path (radiodsp_sdr_rx_amd/) never links or calls it.
 *
 * PARITY PINNED on the reference's own compiled code for the stages its shipped build contains (the CONV
 * stage with its NLMS, the filter design, both integer analysers, AudioFilterBiquad, the CMSIS routines under
 * them: known answers made by running the firmware image's routines under an instruction-set interpreter,
 * tests/golden/firmware_kat.npz, tests/test_firmware_kat.py) and on its constant tables
 * (tests/test_firmware_tables.py); PARITY UNPINNED for the stages that are not in that build or not in the
 * tree (decimator, spectral stage, the AudioSDR engine's NCO / ALS / AGC): see the .c header.  The reference
 * (gcallipo/RadioDSP_SDR_RX, an Arduino/Teensy sketch) ships no tests, golden vectors or fixtures, and cannot
 * be compiled here (needs the Arduino core, Teensy Audio library, CMSIS-DSP and the AudioSDR library, none of
 * them vendored or pinned).  This file is therefore
 * a plain-C restatement of the in-tree algorithm, following the reference
 * line by line where the code exists:
 *
 *   src/RadioDSP_SDR_RX/RDSP_convolutional.h   (CONV)  filter design, mask,
 *                                                      overlap-save processing
 *   src/RadioDSP_SDR_RX/RDSP_noise_reduction.h (NR)    NLMS noise reduction
 *   src/backup/RDSP_convolutional_spec.h       (SPEC)  spectral subtraction NR
 *
 * and BUILD-DEFINED semantics (documented in DESIGN.md) for the stages that
 * live in the un-vendored AudioSDR library: NCO mixer, polyphase FIR
 * decimator, demodulator selection, LMS auto-notch (ALS filter), AGC.
 * CMSIS-DSP primitives are restated from their published definitions
 * (arm_q15_to_float, arm_float_to_q15, arm_cfft_f32, arm_cmplx_mult_cmplx_f32,
 * arm_cmplx_mag_f32, arm_lms_norm_f32); bit parity with the image's CMSIS holds for all of them but the float
 * FFT (1.2e-7) and arm_cmplx_mag_f32 (not in the image).
 * It is anchored by analytic known-answer tests (tests/test_oracle_*.py) and
 * by an independent float64 NumPy model (tests/np_model.py).
 */
#ifndef RDSP_ORACLE_H
#define RDSP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_BLOCK 128 /* BUFFER_SIZE, CONV:34 */

/* demodulator selection (engine enum names, CTL:330-410) */
enum {
  ORC_DEMOD_IQ = 0,  /* L = Re y, R = Im y : literal CONV:314-318            */
  ORC_DEMOD_USB = 1, /* audio = Re y (one-sided positive passband)            */
  ORC_DEMOD_LSB = 2, /* audio = Re y (one-sided negative passband)            */
  ORC_DEMOD_CW_USB = 3,
  ORC_DEMOD_CW_LSB = 4,
  ORC_DEMOD_AM = 5, /* audio = |y| - block-smoothed DC                       */
  ORC_DEMOD_SAM = 6 /* synchronous AM: PLL on the carrier, audio = Re(y e^-j phi) */
};

enum { ORC_AGC_OFF = 0, ORC_AGC_FAST = 1, ORC_AGC_MEDIUM = 2, ORC_AGC_SLOW = 3 };
enum { ORC_ALS_OFF = 0, ORC_ALS_NOTCH = 1, ORC_ALS_PEAK = 2 };

/* One receiver channel's configuration.  Field order is mirrored by
 * rdsp_chain_config_t in include/rdsp.h (tests fill both from one dict). */
typedef struct {
  double fs_in;        /* input IQ rate, Hz                                   */
  int32_t decim;       /* 1 (no decimator) or 4                               */
  int32_t fir_taps;    /* decimator taps (256); ignored when decim == 1       */
  double fir_cut_hz;   /* decimator low-pass half-width B, Hz                 */
  double nco_hz;       /* tuning offset removed by the mixer (0 = mixer off)  */
  int32_t fft_l;       /* FFT_L: 256/512/1024/2048/4096 (CONV:36)             */
  int32_t window;      /* FIR_filter_window id, 1 = Blackman-Harris (CONV:66) */
  double flo_hz;       /* FLoCut (CONV:67)                                    */
  double fhi_hz;       /* FHiCut (CONV:68)                                    */
  int32_t filter_on;   /* bFilterEnabled (CONV:300)                           */
  int32_t demod;       /* ORC_DEMOD_*                                         */
  int32_t spectral_nr; /* 0 off, 1: spectral subtraction (SPEC:112-269), 2: the older
                          variant of backup/RadioDSP_SDR_RX_Conv.ino:1520-1669        */
  float spectral_level;/* iNRLevel of SPEC:112 (0..3)                         */
  int32_t lms_nr;      /* 0 = off, else DSP-NR strength (NR:35; 15,20..50)    */
  int32_t als_mode;    /* ORC_ALS_* : LMS auto-notch / peak                   */
  int32_t als_strength;/* strength for the ALS instance's mu (same law)      */
  int32_t agc_mode;    /* ORC_AGC_*                                           */
  float input_gain;    /* SDR.setInputGain (INO:133)                          */
  float output_gain;   /* SDR.setOutputGain (INO:134)                         */
  float iq_balance;    /* SDR.setIQgainBalance (INO:135): I *= g              */
  int32_t mute;        /* SDR.setMute                                         */
} orc_config_t;

typedef struct orc_chain orc_chain_t;

/* ---- CMSIS-DSP primitive restatements ---------------------- */
void orc_q15_to_float(const int16_t *src, float *dst, uint32_t n);
void orc_float_to_q15(const float *src, int16_t *dst, uint32_t n);
void orc_cfft_f32(float *buf, uint32_t n, int inverse);
void orc_cmplx_mult_cmplx_f32(const float *a, const float *b, float *dst, uint32_t n);
void orc_cmplx_mag_f32(const float *src, float *dst, uint32_t n);

/* ---- CONV:127-185 / CONV:87-110 ----------------------------------------- */
void orc_calc_cplx_FIR_coeffs(double *coeffs_I, double *coeffs_Q, int numCoeffs,
                              double FLoCut, double FHiCut, double SampleRate,
                              int window);
/* mask must hold 2*fft_l floats; taps: m_NumTaps = fft_l/2+1 doubles each */
void orc_init_filter_mask(float *mask, const double *coef_I, const double *coef_Q,
                          uint32_t fft_l);

/* ---- chain object --------------------------------------------------------*/
orc_chain_t *orc_chain_create(const orc_config_t *cfg);
void orc_chain_destroy(orc_chain_t *c);

/* reference-named stage calls, with the context pointer the globals became */
void orc_doConvolutionalInitialize(orc_chain_t *c);            /* CONV:187 */
void orc_reInitializeFilter(orc_chain_t *c, double lo, double hi); /* CONV:209 */
void orc_Init_LMS_NR(orc_chain_t *c, int strength);            /* NR:35    */
void orc_LMS_NoiseReduction(orc_chain_t *c, int16_t n, float *nrbuffer); /* NR:66 */
/* arm_lms_norm_f32 alone (96 taps) on an instance given as arrays: coeffs[96], state95[95], energy_x0[2] in/out */
void orc_lms_norm_f32_kat(float mu, float *coeffs, float *state95, float *energy_x0, const float *src, const float *ref,
                          float *out, float *err, uint32_t n);
void orc_Init_ALS(orc_chain_t *c, int strength);
void orc_set_nr_level(orc_chain_t *c, int lms_nr);  /* nr_level change, CONV:327 */

/* F3 (build-defined engine features): preProcessor.swapIQ (INO:118), noise blanker
 * (BK_INO:1259-1260, INO:131) */
void orc_set_swap_iq(orc_chain_t *c, int on);
void orc_set_iq_slip(orc_chain_t *c, int slip); /* INO:117: +1 delays the I rail by one sample, -1 the Q rail */
void orc_set_noise_blanker(orc_chain_t *c, int on, float threshold_db);
float orc_chain_nb_level(const orc_chain_t *c);
void orc_set_gains(orc_chain_t *c, float input_gain, float iq_balance, float output_gain, int mute);
void orc_set_agc_mode(orc_chain_t *c, int mode);
void orc_set_filter_on(orc_chain_t *c, int on);
void orc_set_als_mode(orc_chain_t *c, int mode);
void orc_set_spectral_nr(orc_chain_t *c, int on, float level);
/* SPEC:221-235 as written (atan2 + table-interpolated arm_cos_f32 / arm_sin_f32) instead of the equivalent
 * X * mag'/mag: bounds how far the two forms are apart (test infrastructure) */
void orc_set_literal_resynthesis(orc_chain_t *c, int on);
/* CONV:303 as written (filter off copies half the spectrum, the rest of iFFT_buffer stale) instead of the full
 * bypass the restatement uses: shows what that choice replaces (test infrastructure) */
void orc_set_literal_filter_off(orc_chain_t *c, int on);
void orc_set_literal_nr_first_block(orc_chain_t *c, int on); /* CONV:326-337 as written for N_BLOCKS > 1 */
float orc_arm_sin_f32(float x);
void orc_arm_sin_table(float *tab513); /* sinTable_f32 as published (eight-place decimal literals) */
float orc_arm_cos_f32(float x);
/* SAM PLL loop constants at the decimated rate (build-defined, see rdsp_oracle.c) */
void orc_sam_constants(double fs_out, float *g1, float *g2, float *wmin, float *wmax);

/* F2: retune / PBT / mode table -- the callers of reInitializeFilter (CTL:569-612)
 * and the mode menu (CTL:330-423) */
void orc_set_demod(orc_chain_t *c, int demod);
void orc_set_nco_hz(orc_chain_t *c, double hz);
void orc_pbt_step(double *dFLoCut, double *dFHiCut, int edge, int dir);
void orc_passband(int filter, int demod, double *lo, double *hi);
int orc_tuning_mode(int mndx, double vfoFreq, int *filter, int *demod);

/* Process n_blocks blocks of 128 interleaved int16 IQ samples.  Output is
 * produced hop by hop (hop = fft_l/2 samples at the decimated rate); returns
 * the number of output sample pairs written.  out_i16 is interleaved L,R
 * (arm_float_to_q15 of the float audio); out_f32 (optional) receives the
 * float L,R pairs before packing. */
int orc_chain_process(orc_chain_t *c, const int16_t *iq, int n_blocks,
                      int16_t *out_i16, float *out_f32);

/* read-only views for tests */
const float *orc_chain_mask(const orc_chain_t *c);
const float *orc_chain_fir_taps(const orc_chain_t *c);
const float *orc_chain_lms_coeffs(const orc_chain_t *c, int which);
float orc_chain_nfloor(const orc_chain_t *c);
float orc_chain_agc_gain(const orc_chain_t *c);
uint32_t orc_chain_nco_dphi(const orc_chain_t *c);
/* build-defined: tuning offset per demod mode, Hz (setDemodMode return) */
uint32_t orc_demod_tuning_offset(int demod);

/* Many-channel convenience used by the cpu_baseline leg: runs n_ch
 * independent chains (same config) over iq[ch][n_blocks*128][2] with
 * n_threads OpenMP threads.  Returns output pairs per channel. */
int orc_multi_process(const orc_config_t *cfg, int n_ch, const int16_t *iq,
                      int n_blocks, int16_t *out_i16, int n_threads);

#ifdef __cplusplus
}
#endif
#endif

/* ---- F1: IQ panadapter spectrum (analyze_fft256iq.{h,cpp}) -----------------------
 * Restatement of AudioAnalyzeFFT256IQ::update (FFTIQ.cpp:65-118).  The pieces that live
 * outside the tree follow the libraries' published routines with the tables of the
 * reference's firmware image: q15 windows (Teensy windows.c), arm_cfft_radix4_q15 (CMSIS,
 * DSP-extension form, twiddleCoef_4096_q15) and sqrt_uint32_approx (Teensy
 * utility/sqrt_integer.h, guess table) -- see oracle/rdsp_oracle.c. */
#ifndef RDSP_ORACLE_SPECTRUM
#define RDSP_ORACLE_SPECTRUM
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_fft256iq orc_fft256iq_t;
void orc_window_q15(int window_id, int16_t *w256);           /* ids: see orc_window_q15_n in the .c file */
void orc_cfft_radix4_q15_256(int16_t *buf /* 512: re,im */); /* scaled by 1/256, natural order */
void orc_twiddle_q15_4096(int16_t *table6144);               /* twiddleCoef_4096_q15 by its rule */
uint32_t orc_sqrt_uint32(uint32_t x);                        /* exact floor root (comparison only) */
/* sqrt_uint32_approx (FFTIQ.cpp:105): Teensy Audio's published routine with the firmware image's guess table */
uint32_t orc_sqrt_uint32_approx(uint32_t in);
const uint16_t *orc_sqrt_guess_table(void);
orc_fft256iq_t *orc_fft256iq_create(int naverage, int window_id);
void orc_fft256iq_destroy(orc_fft256iq_t *s);
void orc_fft256iq_averageTogether(orc_fft256iq_t *s, int n);       /* FFTIQ.h:88-91 */
void orc_fft256iq_windowFunction(orc_fft256iq_t *s, int window_id); /* FFTIQ.h:93-95 */
void orc_fft256iq_windowFunction_table(orc_fft256iq_t *s, const int16_t *w256); /* the same, the reference's signature */
float orc_fft256iq_read(const orc_fft256iq_t *s, unsigned int binNumber);       /* FFTIQ.h:70-73 */
float orc_fft256iq_read_range(const orc_fft256iq_t *s, unsigned int binFirst, unsigned int binLast); /* FFTIQ.h:75-86 */
/* one update() tick with a 128-sample I block and Q block; returns 1 when output[] was refreshed */
int orc_fft256iq_update(orc_fft256iq_t *s, const int16_t *block_i, const int16_t *block_q);
const uint16_t *orc_fft256iq_output(const orc_fft256iq_t *s); /* uint16 output[256], FFTIQ.h:99 */
/* AudioAnalyzeFFT1024 (Teensy Audio library; INO:57,87), restated like F1 */
typedef struct orc_fft1024 orc_fft1024_t;
void orc_window_q15_n(int window_id, int n, int16_t *w);
void orc_cfft_radix4_q15_n(int16_t *buf, int n);
orc_fft1024_t *orc_fft1024_create(int window_id);
void orc_fft1024_destroy(orc_fft1024_t *s);
void orc_fft1024_windowFunction_table(orc_fft1024_t *s, const int16_t *w1024);
int orc_fft1024_update(orc_fft1024_t *s, const int16_t *block);
const uint16_t *orc_fft1024_output(const orc_fft1024_t *s); /* uint16 output[512] */
/* biquad cascade of the engine's audio filter bank (arm_biquad_cascade_df1_f32): DF1, float, CMSIS coefficient
 * order {b0, b1, b2, a1, a2} per stage with the feedback terms added */
#define ORC_BIQUAD_MAX_STAGES 4
typedef struct { int n_stages; float coef[5 * ORC_BIQUAD_MAX_STAGES]; float state[4 * ORC_BIQUAD_MAX_STAGES]; } orc_biquad_t;
void orc_biquad_init(orc_biquad_t *b, int n_stages, const float *coef5);
void orc_biquad_set_stage(orc_biquad_t *b, int stage, const float *coef5);
void orc_biquad_run(orc_biquad_t *b, float *x, int n);
void orc_biquad_design(int kind, double freq, double q, double fs, float *coef5); /* 0 LP 1 HP 2 BP 3 notch */
/* AudioFilterBiquad of the Teensy Audio library as published: fixed-point cascade, coefficients x 2^30, SMLAWB/T
 * products, 14-bit error feedback, SSAT >> 14 (see the .c file) */
typedef struct { int chained[4]; int32_t coef[4][5]; int16_t x1[4], x2[4], y1[4], y2[4]; int32_t sum[4]; } orc_teensy_biquad_t;
void orc_teensy_biquad_init(orc_teensy_biquad_t *b);
void orc_teensy_biquad_setCoefficients_int(orc_teensy_biquad_t *b, int stage, const int32_t *coef5);
void orc_teensy_biquad_setCoefficients(orc_teensy_biquad_t *b, int stage, const double *coef5);
void orc_teensy_biquad_design(int kind, float frequency, float q, float fs, int32_t *coef5);
void orc_teensy_biquad_update(orc_teensy_biquad_t *b, int16_t *data, int n);
void orc_design_butter_bp8(double f1, double f2, double fs, float *coef20);
void orc_set_audio_iir(orc_chain_t *c, int on, double f1, double f2);
const float *orc_chain_iir_coeffs(const orc_chain_t *c);
uint64_t orc_fft256iq_multi(int naverage, int window_id, int n_ch, const int16_t *iq, int n_blocks, int n_threads);
#ifdef __cplusplus
}
#endif
/* ---- the reference's AudioSDR engine as its firmware image computes it (oracle/rdsp_engine_oracle.c) ---------------- */
typedef struct orc_engine orc_engine_t;
float orc_newlib_expf(float x);
orc_engine_t *orc_engine_create(const float *biquad_sets15x20, const float *hilbert64);
void orc_engine_destroy(orc_engine_t *e);
void orc_engine_update(orc_engine_t *e, const int16_t *i128, const int16_t *q128, int16_t *out128); /* image 0xe730 */
float orc_engine_setDemodMode(orc_engine_t *e, int mode);   /* INO:139, CTL:337-407 */
void orc_engine_setAudioFilter(orc_engine_t *e, int id);    /* INO:138, CTL:153-177 */
void orc_engine_enableAudioFilter(orc_engine_t *e);         /* INO:137 */
void orc_engine_setInputGain(orc_engine_t *e, float g);     /* INO:133 */
void orc_engine_setOutputGain(orc_engine_t *e, float g);    /* INO:134 */
void orc_engine_setIQgainBalance(orc_engine_t *e, float b); /* INO:135 */
void orc_engine_setMute(orc_engine_t *e, int on);           /* INO:177 */
void orc_engine_enableAGC(orc_engine_t *e);                 /* INO:120 */
void orc_engine_setAGCmode(orc_engine_t *e, int mode);      /* INO:121, CTL:200-218 */
void orc_engine_enableALSfilter(orc_engine_t *e);           /* CTL:259 */
void orc_engine_disableALSfilter(orc_engine_t *e);          /* INO:125 */
void orc_engine_setALSfilterNotch(orc_engine_t *e);         /* CTL:260 */
void orc_engine_setALSfilterPeak(orc_engine_t *e);
void orc_engine_setALSfilterAdaptive(orc_engine_t *e);      /* CTL:261 */
void orc_engine_enableNoiseBlanker(orc_engine_t *e);
void orc_engine_disableNoiseBlanker(orc_engine_t *e);       /* INO:131 */
void orc_engine_set_tap(orc_engine_t *e, float *buf9x2x128);
float orc_engine_scalar(const orc_engine_t *e, int which);
const float *orc_engine_agc_curve(const orc_engine_t *e);
const float *orc_engine_sine(const orc_engine_t *e);
const float *orc_engine_als_taps(const orc_engine_t *e);

/* AudioSDRpreProcessor, INO:53,117-118 (image 0xee88, 0xf084) */
typedef struct orc_preproc orc_preproc_t;
orc_preproc_t *orc_preproc_create(void);
void orc_preproc_destroy(orc_preproc_t *p);
void orc_preproc_startAutoI2SerrorDetection(orc_preproc_t *p);
void orc_preproc_swapIQ(orc_preproc_t *p, int on);
int orc_preproc_state(const orc_preproc_t *p, int which); /* 0 slip (-1, 0, 1), 1 bad count, 2 checked blocks, 3 detecting */
void orc_preproc_update(orc_preproc_t *p, int16_t *i128, int16_t *q128);

#endif
