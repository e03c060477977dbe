/*
 * rdsp.h -- C-ABI of the MI355X-native many-channel SDR receive chain.
 *
 * Drop-in boundary for the per-block IQ receive path of gcallipo/RadioDSP_SDR_RX.
 * Every entry point cites the reference interface it replaces (paths relative
 * to /root/reference/src):
 *   CONV = RadioDSP_SDR_RX/RDSP_convolutional.h   NR  = RadioDSP_SDR_RX/RDSP_noise_reduction.h
 *   SPEC = backup/RDSP_convolutional_spec.h       INO = RadioDSP_SDR_RX/RadioDSP_SDR_RX.ino
 *   CTL  = RadioDSP_SDR_RX/RDSP_controls.h        FFTIQ = RadioDSP_SDR_RX/analyze_fft256iq.{h,cpp}
 *
 * The reference keeps all DSP state in single-instance globals (CONV:34-80,
 * NR:18-32); here the same functions take a context pointer, and one context
 * (`rdsp_chain_t`) holds n_channels independent receivers on one GPU.  Plain C:
 * pointers and sizes only.  `d_` pointers are device (HBM) addresses, `stream`
 * is a hipStream_t passed as void*.  All functions returning int return
 * RDSP_OK (0) or a negative RDSP_ERR_*; rdsp_last_error() gives the text.
 * There is no CPU fallback: without a GPU every compute call fails loudly.
 */
#ifndef RDSP_H
#define RDSP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RDSP_BLOCK_SAMPLES 128 /* AUDIO_BLOCK_SAMPLES / BUFFER_SIZE, CONV:34, FFTIQ.cpp:44 */

enum {
  RDSP_OK = 0,
  RDSP_ERR_INVALID = -1,     /* bad argument / unsupported configuration */
  RDSP_ERR_NO_DEVICE = -2,   /* no HIP device: the product has no CPU path */
  RDSP_ERR_HIP = -3,         /* a HIP runtime call failed */
  RDSP_ERR_NOT_READY = -4,   /* not enough queued input (CONV:231) */
  RDSP_ERR_UNSUPPORTED = -5, /* declared engine feature not built yet */
  RDSP_ERR_NOMEM = -6
};

/* demodulator modes: engine enum of CTL:330-410 (+ IQ = literal CONV:314-318) */
typedef enum {
  RDSP_DEMOD_IQ = 0,
  RDSP_DEMOD_USB = 1,    /* USBmode    */
  RDSP_DEMOD_LSB = 2,    /* LSBmode    */
  RDSP_DEMOD_CW_USB = 3, /* CW_USBmode */
  RDSP_DEMOD_CW_LSB = 4, /* CW_LSBmode */
  RDSP_DEMOD_AM = 5,     /* AMmode     */
  RDSP_DEMOD_SAM = 6     /* SAMmode (CTL:387): PLL synchronous detector, build-defined */
} rdsp_demod_t;
typedef enum { RDSP_AGC_OFF = 0, RDSP_AGC_FAST = 1, RDSP_AGC_MEDIUM = 2, RDSP_AGC_SLOW = 3 } rdsp_agc_t; /* CTL:200-218 */
typedef enum { RDSP_ALS_OFF = 0, RDSP_ALS_NOTCH = 1, RDSP_ALS_PEAK = 2 } rdsp_als_t;                    /* CTL:250-261 */
/* SDR.setAudioFilter enum, CTL:153-177,396 */
typedef enum { RDSP_AUDIO_CW = 0, RDSP_AUDIO_2100 = 1, RDSP_AUDIO_2700 = 2, RDSP_AUDIO_3100 = 3,
               RDSP_AUDIO_AM = 4, RDSP_AUDIO_WSPR = 5 } rdsp_audio_filter_t;

/* One configuration shared by all channels of a chain (the reference's
 * compile-time constants CONV:34-38,66-72 and run-time globals GEN:76-111). */
typedef struct {
  double fs_in;         /* input IQ rate, Hz (AUDIO_SAMPLE_RATE_EXACT role, CONV:35) */
  int32_t decim;        /* 1 = no decimator (reference-native), 4 = polyphase /4   */
  int32_t fir_taps;     /* decimator taps, 256                                       */
  double fir_cut_hz;    /* decimator low-pass half-width, Hz                         */
  double nco_hz;        /* tuning offset removed by the mixer (TuningOffset, INO:139) */
  int32_t fft_l;        /* FFT_L, CONV:36: 256 512 1024 2048 4096                    */
  int32_t window;       /* FIR_filter_window, CONV:66 (1 = Blackman-Harris)          */
  double flo_hz;        /* FLoCut, CONV:67                                           */
  double fhi_hz;        /* FHiCut, CONV:68                                           */
  int32_t filter_on;    /* bFilterEnabled, CONV:228,300                              */
  int32_t demod;        /* rdsp_demod_t                                              */
  int32_t spectral_nr;  /* spectral-subtraction NR: 0 off, 1 SPEC:112-269, 2 the older
                           variant of backup/RadioDSP_SDR_RX_Conv.ino:1520-1669 (bins
                           60..120, /60, x3, no smoothing; the sketch runs it at nrndx == 3) */
  float spectral_level; /* iNRLevel of SPEC:112,202                                  */
  int32_t lms_nr;       /* nr_level: 0 off, else DSP-NR strength, CONV:326, NR:35    */
  int32_t als_mode;     /* rdsp_als_t                                                */
  int32_t als_strength; /* mu law input for the ALS instance                         */
  int32_t agc_mode;     /* rdsp_agc_t                                                */
  float input_gain;     /* SDR.setInputGain, INO:133                                 */
  float output_gain;    /* SDR.setOutputGain, INO:134                                */
  float iq_balance;     /* SDR.setIQgainBalance, INO:135                             */
  int32_t mute;         /* SDR.setMute, INO:177                                      */
} rdsp_chain_config_t;

typedef struct rdsp_chain rdsp_chain_t;

const char *rdsp_last_error(void);
const char *rdsp_version(void);
/* 1 if the library was built with EXPERIMENTAL=1 (csrc/Makefile): it then also carries the
 * kernel variants that were measured and not adopted (matrix-core FIR and tail reductions,
 * half-row / DPP-shift tail layouts); the product build has none of them and the variant
 * setters below return RDSP_ERR_UNSUPPORTED for anything but the defaults. */
int rdsp_experimental_build(void);
int rdsp_device_count(void);

/* ---- host-side design helpers ---------------------------------------------*/
/* calc_cplx_FIR_coeffs, CONV:127 (window id added as a parameter: the
 * reference reads the global FIR_filter_window, CONV:66) */
void rdsp_calc_cplx_FIR_coeffs(double *coeffs_I, double *coeffs_Q, int numCoeffs, double FLoCut,
                               double FHiCut, double SampleRate, int window);
/* init_filter_mask, CONV:87: natural-order mask, 2*fft_l floats */
int rdsp_init_filter_mask(float *mask, const double *coef_I, const double *coef_Q, int fft_l);

/* ---- chain: n_channels receivers on one GPU ---------------------------------*/
/* max_blocks_per_call sizes the intermediate audio buffer (no allocation ever
 * happens inside rdsp_chain_process). */
int rdsp_chain_create(const rdsp_chain_config_t *cfg, int n_channels, int device,
                      int max_blocks_per_call, rdsp_chain_t **out);
void rdsp_chain_destroy(rdsp_chain_t *c);
int rdsp_chain_channels(const rdsp_chain_t *c);
/* number of 128-sample input blocks per call must be a multiple of this
 * (N_BLOCKS of CONV:38-39 seen from the input rate: 8 for FFT_L <= 512 at decim 4) */
int rdsp_chain_call_unit_blocks(const rdsp_chain_t *c);
/* The unit a stream is cut in for output bits that do not depend on the cut.  The call unit for the default
 * decimator, the direct form and decim-1 chains (any call split gives the same bits).  With the 448-sample decimator
 * frames of rdsp_chain_set_fir_variant(c, 2) -- 14 input blocks a frame -- lcm(14, call unit): 56 blocks for
 * FFT_L <= 512.  Calls of whole granules are whole frames, the frame grid then sits at absolute stream positions,
 * and the stream gives the same bits however it is cut into such calls: restricting the call boundaries is how the
 * reference has the property too (`available() > N_BLOCKS`, CONV:231).  Other multiples of the call unit are
 * accepted in that form; such a call ends in a partial frame and re-anchors the grid at the next call's first
 * sample (valid output, rounds differently: <= 3e-7 of the peak). */
int rdsp_chain_granule_blocks(const rdsp_chain_t *c);
/* zero all per-channel state (overlap block, FIR history, NLMS, NFloor, AGC) */
int rdsp_chain_reset(rdsp_chain_t *c, void *stream);

/* doConvolutionalInitialize(), CONV:187 */
int rdsp_doConvolutionalInitialize(rdsp_chain_t *c, void *stream);
/* reInitializeFilter(lo, hi), CONV:209 (PBT retune path CTL:569-612) */
int rdsp_reInitializeFilter(rdsp_chain_t *c, double dFLoCut, double dFHiCut, void *stream);
/* Init_LMS_NR(strength), NR:35 */
int rdsp_Init_LMS_NR(rdsp_chain_t *c, int LMS_nr_strength, void *stream);

/* LMS_NoiseReduction(blockSize, nrbuffer), NR:66: the DSP-NR NLMS instance
 * alone, in place on float audio d_nrbuffer[n_channels][stride]; n_samples per
 * channel, a multiple of 128 (the reference calls it with 128, CONV:332) */
int rdsp_LMS_NoiseReduction(rdsp_chain_t *c, int n_samples, float *d_nrbuffer, size_t stride,
                            void *stream);

/* The hot path.  doConvolutionalProcessing (CONV:228 / SPEC:112) with the
 * engine stages in front and behind it, for every channel, over n_blocks
 * input blocks of 128 IQ samples per channel.
 *   d_iq   : int16 [n_channels][in_stride][2]   (I, Q interleaved)
 *   d_out  : int16 [n_channels][out_stride][2]  (L, R interleaved), receives
 *            n_blocks*128/decim sample pairs per channel
 *   d_out_f32 : optional float [n_channels][out_stride][2], the same audio
 *            before arm_float_to_q15 (parity tests)
 * strides are in samples (pairs); in_stride must be a multiple of 4. */
int rdsp_chain_process(rdsp_chain_t *c, const int16_t *d_iq, size_t in_stride, int n_blocks,
                       int16_t *d_out, size_t out_stride, float *d_out_f32, void *stream);
/* reference-shaped call: nr level and filter enable per call (CONV:228);
 * the cut-off arguments are ignored exactly as in the reference (CONV:300). */
int rdsp_doConvolutionalProcessing(rdsp_chain_t *c, float iNRLevel, int bFilterEnabled,
                                   double dFLoCut, double dFHiCut, const int16_t *d_iq,
                                   size_t in_stride, int n_blocks, int16_t *d_out,
                                   size_t out_stride, void *stream);

/* arm_q15_to_float / arm_float_to_q15 call sites CONV:241-242,346-347.  q / 32768.0f; and, in the variant the reference's
 * firmware image holds (CMSIS under ARM_MATH_ROUNDING): in = x * 32768, in += in > 0 ? 0.5f : -0.5f, (q15_t)__SSAT((q31_t)in,
 * 16) -- round to nearest, halves away from zero, saturating; finite input.  Every int16 output of the library is packed
 * this way. */
int rdsp_q15_to_float(const int16_t *d_src, float *d_dst, size_t n, void *stream);
int rdsp_float_to_q15(const float *d_src, int16_t *d_dst, size_t n, void *stream);

/* ---- engine setters: AudioSDR API visible at INO:117-139, CTL:149-423 ------*/
int rdsp_sdr_enableAGC(rdsp_chain_t *c);                          /* INO:120 */
int rdsp_sdr_disableAGC(rdsp_chain_t *c);                         /* CTL:268 */
int rdsp_sdr_setAGCmode(rdsp_chain_t *c, int mode);               /* INO:121, CTL:200-218 */
int rdsp_sdr_enableALSfilter(rdsp_chain_t *c);                    /* CTL:259 */
int rdsp_sdr_disableALSfilter(rdsp_chain_t *c);                   /* INO:125 */
int rdsp_sdr_setALSfilterNotch(rdsp_chain_t *c);                  /* CTL:260 */
int rdsp_sdr_setALSfilterPeak(rdsp_chain_t *c);                   /* BK_INO:665 */
int rdsp_sdr_setALSfilterAdaptive(rdsp_chain_t *c);               /* CTL:261 */
/* noise blanker: engine feature, arithmetic build-defined: on the
 * wide-band IQ stream before the mixer, a sample whose power exceeds the reference
 * level by threshold dB (default 10, range 0..60) is zeroed; the reference level is
 * the smoothed mean power of the previous windows of 256*decim input samples */
int rdsp_sdr_enableNoiseBlanker(rdsp_chain_t *c);                 /* BK_INO:1259 */
int rdsp_sdr_disableNoiseBlanker(rdsp_chain_t *c);                /* INO:131 */
int rdsp_sdr_setNoiseBlankerThresholdDb(rdsp_chain_t *c, float db); /* BK_INO:1260 */
/* AudioSDRpreProcessor */
/* swapIQ and the input-side gains take effect with the first sample of the next call (samples
 * already inside the decimator's delay line keep what they came in with) */
int rdsp_pre_swapIQ(rdsp_chain_t *c, int swap);                   /* INO:118 */
int rdsp_pre_startAutoI2SerrorDetection(rdsp_chain_t *c);         /* INO:117: accepted; there is no I2S bus to watch at run time */
/* What INO:117 guards against is an I2S fault that leaves one rail of the codec stream a sample behind
 * the other; a RECORDING made through such a front end carries it (the image rejection of the
 * quadrature pair is gone).  rdsp_estimate_iq_slip finds it in a recording (host, no GPU),
 * rdsp_pre_setIQslip corrects it: slip +1 pairs I[n-1] with Q[n] (delays the I rail by one sample),
 * -1 pairs I[n] with Q[n-1], 0 switches the correction off.  It acts on the raw words in a pass of
 * its own in front of the front kernel (8 bytes of HBM traffic per input sample while it is on), before
 * swapIQ and the input gains, on samples as they arrive from the next call on.  The first non-zero
 * value allocates the corrected-input buffer (n_channels x max_blocks_per_call x 128 words): a set-up
 * call.  Build-defined (the AudioSDR pre-processor is not in the reference tree). */
int rdsp_pre_setIQslip(rdsp_chain_t *c, int slip);
/* iq: n_samples interleaved int16 I,Q pairs of ONE channel (host memory).  *slip: the value to pass to
 * rdsp_pre_setIQslip; rejection_db (optional, 3 values): image rejection of the strongest line with the
 * slip undone as 0, +1, -1.  Needs a dominant one-sided line in the first 2^k samples (k <= 14); without
 * one (noise, a real-valued or double-side-band signal: all three rejections near 0 dB) *slip is 0 -- a
 * correction is only recommended when it rejects the image by 15 dB at least and beats "no slip" by 10 dB. */
int rdsp_estimate_iq_slip(const int16_t *iq, size_t n_samples, int *slip, double *rejection_db);
int rdsp_sdr_setInputGain(rdsp_chain_t *c, float g);              /* INO:133 */
int rdsp_sdr_setOutputGain(rdsp_chain_t *c, float g);             /* INO:134 */
int rdsp_sdr_setIQgainBalance(rdsp_chain_t *c, float g);          /* INO:135 */
int rdsp_sdr_enableAudioFilter(rdsp_chain_t *c);                  /* INO:137 */
int rdsp_sdr_setAudioFilter(rdsp_chain_t *c, int filter, void *stream); /* INO:138, CTL:153-177 */
/* returns the tuning offset in Hz like the reference (INO:139, CTL:337-407) -- the AudioSDR engine's own answers, read
 * by running its constructor and setDemodMode out of the reference's firmware image (tests/test_firmware_kat.py): a low
 * IF centred on 6890 Hz, the carrier half a band (SSB 3000 Hz, CW 1000 Hz) above it for the lower and below it for the
 * upper side band: LSBmode 8390, USBmode 5390, CW_LSBmode 7390, CW_USBmode 6390, AMmode / SAMmode 6890 (RDSP_DEMOD_IQ,
 * the literal CONV stage: 0).  The sketch puts its LO at vfoFreq - TuningOffset (CTL:447) and the engine moves the
 * carrier from there to 0 Hz itself; in this library the mixer is a setting of its own, so a host that mirrors the
 * sketch hands the returned value to rdsp_sdr_setTuningOffsetHz (tests/host/rdsp_binding.h does), and a host with many
 * carriers in one recorded stream sets each group's mixer to where its carrier is.  The engine's modes pick
 * the side band the selected audio filter sits on, so the pass band is re-applied for the new mode
 * (USB/CW_USB: +a..+b, LSB/CW_LSB: -b..-a, AM/SAM: -b..+b of setAudioFilter's a..b); a later
 * rdsp_reInitializeFilter / PBT step overrides it, as RDSP_controls.h:569-612 does in the sketch. */
uint32_t rdsp_sdr_setDemodMode(rdsp_chain_t *c, int mode, void *stream);
int rdsp_sdr_setMute(rdsp_chain_t *c, int mute);                  /* INO:177 */
int rdsp_sdr_setTuningOffsetHz(rdsp_chain_t *c, double hz);       /* NCO side of CTL:447 */
int rdsp_set_nr_level(rdsp_chain_t *c, int nr_level);             /* nr_level, GEN:111, CTL:237-297 */
/* The window energy of arm_lms_norm_f32 (NR:73) in both NLMS instances.  running = 0 (default): the reference's running
 * difference, re-started from the exact 96-sample window sum at every 128-sample block -- a deviation from NR:73, made
 * because the reference's own float32 recursion leaves `energy + 1.19e-7 <= 0` after loud-to-quiet transitions (164 of
 * 320 in tests/test_gpu_parity.py) and can lose a channel for good; on ordinary signals the two agree to the reference's
 * accumulated rounding (~1e-6).  running = 1: no anchor -- one running sum for the whole stream like NR:73's, but in the
 * kernel's own form: the increments are fused x^2 - q^2 terms added by a 16-lane prefix scan, where arm_lms_norm_f32
 * subtracts and adds sample by sample.  Neither bit-equal to the reference's recursion nor statistically the same (the
 * scan form draws bad residues more often: 90 channels flagged / 8 lost on the loud-to-quiet test against 1 in the CPU
 * restatement); tested <= 1e-5 against the oracle, which always runs the reference's form, and 4.0e-4 (against 1.3e-4 in
 * the default mode) from the firmware image's own output on the `conv_fade` fixture. */
int rdsp_set_nlms_energy_mode(rdsp_chain_t *c, int running);
int rdsp_set_spectral_nr(rdsp_chain_t *c, int on, float level);   /* on: 0, 1 (SPEC:112 iNRLevel), 2 (older variant, level unused) */
/* How the spectral stage rebuilds a bin from its new magnitude (SPEC:221-235).  literal = 1: as the file writes it,
 * `mag' * arm_cos_f32(phi)`, `mag' * arm_sin_f32(phi)` with `phi = atan2(im, re)` -- CMSIS-DSP's table sine (513
 * entries, linear interpolation) restated from the published routine.  literal = 0 (default): the exact-arithmetic
 * equivalent X mag'/mag.  The two sit 1.7e-5 ... 1.9e-5 of the output's peak apart (the table's interpolation error,
 * (2 pi / 512)^2 / 8), so "within 1e-5 of the reference's CPU path" can only hold for one of them at a time: each form
 * is tested <= 1e-5 against the oracle evaluating the SAME form (tests/test_gpu_parity.py).  literal = 1 evaluates
 * the table's interpolation in closed form (csrc/rdsp_kernels.hip spec_table_factor: the as-written bin is the exact one
 * times 1 - (h^2 / 2) f (1 - f), f the fraction between table nodes; within 5e-8 of the table's own arithmetic, +5 % on a
 * K3 step); literal = 2 calls atan2f and looks the table up (round 5's code, +35 %); both are held to the oracle's
 * literal form. */
int rdsp_set_spectral_resynthesis(rdsp_chain_t *c, int literal);

/* ---- receiver groups: per-group retune / PBT / mode tables -------
 * The sketch has one receiver, so one filter mask, one tuning offset and one
 * demodulator (globals of CONV:66-80, CTL:330-423).  A chain of many channels is
 * partitioned into groups that each carry their own; channels of a group share
 * the group's mask (a gather by group index in the kernel).  Every group call is
 * the per-group form of a reference call and is non-blocking for the processing
 * stream: the new mask is designed on the host, uploaded on an internal copy
 * stream into the group's idle mask buffer (masks are double-buffered on the
 * device) and switched in, in stream order, by the next rdsp_chain_process; calls
 * already queued keep the old mask.  The chain-wide calls above
 * (rdsp_reInitializeFilter, rdsp_sdr_setDemodMode, ...) apply to every group. */
int rdsp_chain_set_groups(rdsp_chain_t *c, int n_groups, const uint16_t *group_of_channel);
int rdsp_chain_groups(const rdsp_chain_t *c);
int rdsp_group_reInitializeFilter(rdsp_chain_t *c, int group, double dFLoCut, double dFHiCut,
                                  void *stream);                                  /* CONV:209 */
int rdsp_group_setAudioFilter(rdsp_chain_t *c, int group, int filter, void *stream); /* CTL:153-177 */
uint32_t rdsp_group_setDemodMode(rdsp_chain_t *c, int group, int mode, void *stream); /* CTL:337-407 */
int rdsp_group_setTuningOffsetHz(rdsp_chain_t *c, int group, double hz);          /* CTL:447 */
int rdsp_group_get_mask(rdsp_chain_t *c, int group, float *host_out);
/* checkPBT_Increase / checkPBT_Decrease (CTL:569-612): one 50 Hz step of the low
 * (edge 0, button D3) or high (edge 1, button D6) cut-off, dir = +1 / -1, with the
 * limits MIN_LOW 0, MAX_LOW 700, MIN_HI 800, MAX_HI 4000 (GEN:79-82) and the
 * reference's comparisons.  rdsp_pbt_step is the pure host function; rdsp_group_pbt
 * steps a group's cut-offs and calls reInitializeFilter like CTL:575,582,597,605. */
int rdsp_pbt_step(double *dFLoCut, double *dFHiCut, int edge, int dir);
int rdsp_group_pbt(rdsp_chain_t *c, int group, int edge, int dir, void *stream);
/* tuningMode() (CTL:330-423): mndx 0 "CW N", 1 "CW", 2 "USB", 3 "LSB", 4 "AM",
 * 5 "SAM", 6 "RTTY"; the CW side follows vfoFreq > 10 MHz (CTL:337).  Sets the group's audio filter and
 * demodulator, returns TuningOffset in Hz. */
uint32_t rdsp_group_tuningMode(rdsp_chain_t *c, int group, int mndx, double vfo_hz, void *stream);

/* ---- pipelined mode (streaming throughput) -------------------------------------
 * The NLMS/AGC tail stage is serial in time, so its duration is set by the batch
 * length, not by the channel count.  With pipelining on, the tail stage of call k
 * runs on an internal stream concurrently with the front stage of call k+1 (the
 * intermediate audio lives in three buffers).  rdsp_chain_process then returns with
 * the tail possibly still queued: d_out of a call is complete after the next
 * rdsp_chain_flush(c, stream) on the consuming stream. */
int rdsp_chain_set_pipelined(rdsp_chain_t *c, int on);
int rdsp_chain_flush(rdsp_chain_t *c, void *stream);
/* front-kernel variant: -1 auto (full-register), 0 full-register, 1 lean (FFT twiddles
 * rebuilt per pass from one base each: 24 fewer VGPRs, ~3 % slower) */
int rdsp_chain_set_front_variant(rdsp_chain_t *c, int lean);
/* pipelined mode launches a call in channel sub-batches of this many channels (multiple of 64,
 * default 4096, 0 = off) once the chain has at least 1.5 of them: every launch then has the shape
 * the two kernels share a SIMD at, and the sub-batches of one call overlap each other (8192
 * channels: -8 % per step).  Outputs do not depend on it, bit for bit. */
int rdsp_chain_set_sub_batch(rdsp_chain_t *c, int channels);

/* wave priorities in pipelined mode (front kernel during its FIR, tail kernel), 0..3 */
int rdsp_chain_set_priorities(rdsp_chain_t *c, int front_fir_prio, int tail_prio);
/* stage A3 (the decimating FIR, y[m] = sum_k h[k] x[4m - k]; decim 4).  4: in the frequency
 * domain -- polyphase overlap-save: four low-rate transforms, branch spectra, one inverse -- with
 * frames of ONE GRANULE (256 outputs, the rest of the 512-point window zeros): every call boundary is a frame
 * boundary and every frame's input is a function of the absolute sample position, so a stream gives the same
 * bits however it is cut into calls -- the property the reference has by construction (fixed 128-sample
 * blocks, CONV:231-245).  -1 (default): that form, or the row form 5 (below; split-invariant in the same way) for a
 * call that no later stage follows -- no NLMS / ALS / SAM / IIR stage, whose kernel would share the SIMDs -- and
 * that runs without the noise blanker: a function of the chain's settings at the call, never of the call split.
 * 0: the direct form (packed FMAs): split-invariant too, about 1.3x slower.  2: the
 * frequency domain with 448-sample frames: 5 transforms per 448 outputs instead of per 256 (the front kernel
 * ~1.4x faster).  The frames start at the call's first sample, so this form has the property for a restricted
 * set of call boundaries: calls that are multiples of rdsp_chain_granule_blocks() (56 input blocks for FFT_L <=
 * 512) are whole frames and give the same bits for any split into such calls
 * (test_throughput_decimator_is_split_invariant_on_whole_frames); any other multiple of the call unit is accepted
 * and ends in a partial frame -- another split then frames and rounds differently (2.6e-7 of the output's peak
 * over random splits, 2.8e-6 through K3's recursive stages: tests/test_gpu_parity.py pins both).  bench.py selects 2
 * with BASELINE.md's 512-block steps (9 frames and a partial one) and says so in its `config`.  5: the
 * frequency domain on 16-lane rows (rdsp_front_rd_kernel: 256-point windows, four per wave, 128 outputs each, two
 * frames per granule): split-invariant like the default and ~10 % faster than it for chains without a tail stage
 * (K2 0.727 against 0.808 ms per step), no gain beside a tail kernel; with the noise blanker on it runs the default
 * form.  All of them are the same exact linear convolution with the same taps (RDSP_ERR_UNSUPPORTED for 2 / 4 / 5
 * on decim-1 chains, which have no decimator).  EXPERIMENTAL=1 builds: 1 matrix-core GEMM slices, 3 the same unless
 * the tail stage shares the SIMDs, 6 the row form with 192 outputs per window. */
int rdsp_chain_set_fir_variant(rdsp_chain_t *c, int variant);
/* tail-kernel variant: (16, 2) is the product -- a channel per 16-lane DPP row, two
 * steps per DPP reduction, delay line fed from LDS.  EXPERIMENTAL=1 builds: (16, 4) weights one block
 * stale with a hand-interleaved issue order, (16, 3) one reduction per step, (8, 2) half a row per
 * channel, (16 | 8, 1) cross-lane sums on the matrix pipe, (16, 0) delay line shifted by DPP */
int rdsp_chain_set_tail_variant(rdsp_chain_t *c, int lanes_per_channel, int matrix_reduce);

/* ---- per-kernel timing (HIP events on the launch stream; measurement only) ---*/
int rdsp_chain_set_timing(rdsp_chain_t *c, int on);
/* name of the front kernel the most recent call launched, as a profiler shows it without template
 * arguments: "rdsp_front_fd_kernel" (stage A3 in the frequency domain) or "rdsp_front_kernel" */
const char *rdsp_chain_front_kernel_name(const rdsp_chain_t *c);
/* total milliseconds spent in the front and tail kernels over `calls` calls */
int rdsp_chain_get_timing(rdsp_chain_t *c, double *front_ms, double *tail_ms, int *calls);
/* milliseconds between the end of the first and of the last recorded call (calls - 1 steady-state periods
 * of a pipelined sequence: what a long stream pays per call, without the pipeline's fill) */
int rdsp_chain_get_timing_span(rdsp_chain_t *c, double *span_ms, int *calls);

/* ---- state read-back (tests, checkpoint/resume) ------------------------------*/
/* scal: float[n_channels][4] = NFloor (SPEC:109), AGC gain, AM DC, noise-blanker level */
int rdsp_chain_get_scalars(rdsp_chain_t *c, float *host_out, void *stream);
/* which: 0 = DSP-NR instance (NR:31), 1 = ALS instance; float[n_channels][96] in
 * CMSIS coefficient order */
int rdsp_chain_get_lms_coeffs(rdsp_chain_t *c, int which, float *host_out, void *stream);
/* Per-channel health word, host_out[n_channels].  The reference's NLMS keeps its window energy as a
 * running difference (RDSP_noise_reduction.h:73 -> arm_lms_norm_f32): after a loud-to-quiet transition
 * the residue can leave energy + 1.19e-7 at or below zero, the step size turns negative or infinite
 * and the channel's weights run away to +-inf -- the reference's own behaviour, reproduced here.  The
 * tail kernel records it per channel: bits are sticky until rdsp_Init_LMS_NR (DSP-NR instance) or
 * rdsp_chain_reset (both), the arithmetic is untouched.  Synchronises `stream` and the tail stream. */
#define RDSP_STATUS_NR_ENERGY 0x01u     /* DSP-NR: divided by energy + eps <= 0 */
#define RDSP_STATUS_NR_NONFINITE 0x02u  /* DSP-NR: a weight or the energy is inf / NaN */
#define RDSP_STATUS_ALS_ENERGY 0x10u    /* ALS notch / peak instance, likewise */
#define RDSP_STATUS_ALS_NONFINITE 0x20u
int rdsp_chain_get_status(rdsp_chain_t *c, uint32_t *host_out, void *stream);
/* Recovery for the channels the health words name: arm_lms_norm_init_f32 leaves the coefficients
 * (RDSP_noise_reduction.h:62), so Init_LMS_NR does not clear weights that have run away, and the sketch's
 * only cure is a power cycle.  which 0: the DSP-NR instance, 1: the ALS instance; weights, delay block,
 * energy and health word of channels [first_channel, first_channel + n_channels) go back to their boot
 * values, in stream order behind what is queued; every other channel continues bit for bit. */
int rdsp_chain_reset_nlms_channels(rdsp_chain_t *c, int which, int first_channel, int n_channels, void *stream);
/* natural-order filter mask currently in use, float[2*fft_l] (CONV:77) */
int rdsp_chain_get_mask(rdsp_chain_t *c, float *host_out);
int rdsp_chain_get_fir_taps(rdsp_chain_t *c, float *host_out);

/* ---- per-channel state as data (the reference's DSP state is a set of globals,
 * CONV:50-57,77-80, NR:26-32, SPEC:109; here an explicit per-channel record) ------------------------
 * Checkpoint / resume and moving channels between chains or GPUs: the state of channels
 * first_channel .. first_channel + n_channels - 1 (FIR history, overlap block, NFloor / AGC gain / AM DC /
 * blanker level, both NLMS instances, SAM PLL, IIR cascade) as one host blob.  Settings (modes, filters,
 * gains, groups) are configuration and are re-applied by the caller; FFT_L and decimation must match.
 * Loading into a chain that has not processed anything also restores the stream position and call
 * history (resume: the next call continues the stream bit for bit); a chain that has must be at the
 * same stream position.  Control-path calls: both wait for everything queued so far.
 * rdsp_chain_state_bytes is an upper bound for every later rdsp_chain_save_state of that many channels; it
 * grows only with set-up calls that allocate optional state (a SAM group, RDSP_AUDIO_KIND_IIR, the first
 * non-zero rdsp_pre_setIQslip, rdsp_chain_set_groups): size the buffer after set-up.  Blobs carry a
 * version (4 in this library); a blob of another version is refused (RDSP_ERR_INVALID), nothing is restored. */
size_t rdsp_chain_state_bytes(const rdsp_chain_t *c, int n_channels);
int rdsp_chain_save_state(rdsp_chain_t *c, int first_channel, int n_channels, void *host_buf, size_t bytes, void *stream);
int rdsp_chain_load_state(rdsp_chain_t *c, int first_channel, const void *host_buf, size_t bytes, void *stream);

/* ---- recorded IQ in, audio out ----------------------------------
 * The ends of the sketch's graph are an I2S bus and a codec (AudioInputI2S IQinput,
 * AudioOutputI2S audio_out, AudioControlSGTL5000 codec: INO:52,55,159-169).  A
 * host has recordings: little-endian int16 I,Q pairs, RAW (no header) or RIFF/WAVE
 * PCM 16-bit stereo with I on the left channel and Q on the right (the codec wiring
 * INO:71-72); audio goes out as int16 L,R pairs (Q_out_L / Q_out_R, CONV:344-349). */
enum { RDSP_IO_AUTO = 0, RDSP_IO_RAW = 1, RDSP_IO_WAV = 2 };
typedef struct rdsp_iq_reader rdsp_iq_reader_t;
typedef struct rdsp_audio_writer rdsp_audio_writer_t;
int rdsp_iq_reader_open(const char *path, int format, rdsp_iq_reader_t **out);
double rdsp_iq_reader_sample_rate(const rdsp_iq_reader_t *r); /* 0: the container does not say */
int64_t rdsp_iq_reader_frames(const rdsp_iq_reader_t *r);     /* IQ pairs, -1 unknown */
int rdsp_iq_reader_format(const rdsp_iq_reader_t *r);
size_t rdsp_iq_reader_read(rdsp_iq_reader_t *r, int16_t *dst, size_t n_pairs);
void rdsp_iq_reader_close(rdsp_iq_reader_t *r);
int rdsp_audio_writer_open(const char *path, int format, double sample_rate, rdsp_audio_writer_t **out);
size_t rdsp_audio_writer_write(rdsp_audio_writer_t *w, const int16_t *lr, size_t n_pairs);
int64_t rdsp_audio_writer_frames(const rdsp_audio_writer_t *w);
int rdsp_audio_writer_close(rdsp_audio_writer_t *w); /* patches the WAV sizes */

/* Streaming runner: what loop() does with the record/play queues (INO:195-198,
 * CONV:231-244,344-349), for host data.  source fills int16 IQ rows
 * dst[ch * stride_pairs * 2 ...] with n_blocks * 128 pairs per channel and returns
 * the blocks delivered (fewer = end of stream; a trailing partial granule is not
 * processed, as the sketch never processes one); sink receives the int16 L,R rows.
 * Uploads, kernels and downloads of consecutive batches overlap (three streams,
 * two pinned slots).  blocks_per_call must be a multiple of the chain's call unit (of
 * rdsp_chain_granule_blocks() for audio that does not depend on the batch size in every decimator form);
 * max_blocks <= 0 means "until the source ends". */
typedef int (*rdsp_source_fn)(void *user, int16_t *dst, size_t stride_pairs, int n_blocks);
typedef int (*rdsp_sink_fn)(void *user, const int16_t *src, size_t stride_pairs, int n_pairs);
typedef struct {
  int64_t blocks;      /* 128-sample input blocks processed per channel */
  int64_t samples_in;  /* per channel */
  int64_t samples_out; /* per channel */
  double seconds;      /* wall time of the run, set-up included */
  double read_seconds; /* host time inside source() */
  double write_seconds;/* host time inside sink()   */
} rdsp_stream_stats_t;
int rdsp_stream_run(rdsp_chain_t *c, rdsp_source_fn source, void *source_user, rdsp_sink_fn sink,
                    void *sink_user, int blocks_per_call, int64_t max_blocks, rdsp_stream_stats_t *stats);
/* one reader and one writer per channel; the shortest recording ends the run */
int rdsp_stream_run_files(rdsp_chain_t *c, rdsp_iq_reader_t *const *readers,
                          rdsp_audio_writer_t *const *writers, int blocks_per_call, int64_t max_blocks,
                          rdsp_stream_stats_t *stats);
/* host arrays int16 [n_channels][stride_pairs][2] at both ends */
int rdsp_stream_run_memory(rdsp_chain_t *c, const int16_t *host_iq, size_t in_stride_pairs, int64_t n_blocks,
                           int16_t *host_out, size_t out_stride_pairs, int blocks_per_call,
                           rdsp_stream_stats_t *stats);

/* ---- block graph: the AudioStream node/connection API --------------
 * A block is a tile int16 [n_channels][128]; n_channels = 1 is the reference's
 * audio_block_t (FFTIQ.cpp:44,67).  Host-side plumbing in plain C. */
typedef struct rdsp_graph rdsp_graph_t;
typedef struct rdsp_node rdsp_node_t;   /* AudioStream, FFTIQ.h:52-57 */
typedef struct rdsp_block rdsp_block_t; /* audio_block_t */
typedef void (*rdsp_update_fn)(rdsp_node_t *self, void *user); /* virtual void update(void), FFTIQ.h:98 */

rdsp_graph_t *rdsp_graph_create(int n_channels);
void rdsp_graph_destroy(rdsp_graph_t *g);
int rdsp_graph_channels(const rdsp_graph_t *g);
int rdsp_memory(rdsp_graph_t *g, int n_blocks);          /* AudioMemory(40), INO:151 */
int rdsp_memory_usage(const rdsp_graph_t *g);            /* AudioMemoryUsage() */
int rdsp_memory_usage_max(const rdsp_graph_t *g);
/* AudioStream(ninputs, inputQueueArray), FFTIQ.h:55: nodes update in creation order */
rdsp_node_t *rdsp_node_create(rdsp_graph_t *g, int ninputs, rdsp_update_fn update, void *user);
void rdsp_node_set_destructor(rdsp_node_t *n, void (*fn)(void *));
void *rdsp_node_user(rdsp_node_t *n);
rdsp_graph_t *rdsp_node_graph(rdsp_node_t *n);
/* AudioConnection c(src, srcPort, dst, dstPort), INO:71-89 (fan-out allowed) */
int rdsp_connect(rdsp_node_t *src, int src_port, rdsp_node_t *dst, int dst_port);
int rdsp_update_all(rdsp_graph_t *g);                    /* one audio-ISR tick */
void rdsp_no_interrupts(rdsp_graph_t *g);                /* AudioNoInterrupts(), INO:152, CONV:211 */
void rdsp_interrupts(rdsp_graph_t *g);                   /* AudioInterrupts(), INO:175, CONV:222 */
/* inside update() */
rdsp_block_t *rdsp_allocate(rdsp_node_t *n);
rdsp_block_t *rdsp_receive_readonly(rdsp_node_t *n, int port); /* FFTIQ.cpp:70-71 */
rdsp_block_t *rdsp_receive_writable(rdsp_node_t *n, int port);
void rdsp_transmit(rdsp_node_t *n, rdsp_block_t *b, int port);
void rdsp_release(rdsp_block_t *b);                            /* FFTIQ.cpp:114-115 */
int16_t *rdsp_block_data(rdsp_block_t *b);
int rdsp_block_refcount(const rdsp_block_t *b);
/* AudioRecordQueue, CONV:205-206,231-244 */
rdsp_node_t *rdsp_record_queue_create(rdsp_graph_t *g);
void rdsp_record_queue_begin(rdsp_node_t *q);
void rdsp_record_queue_end(rdsp_node_t *q);
int rdsp_record_queue_available(const rdsp_node_t *q);
int16_t *rdsp_record_queue_readBuffer(rdsp_node_t *q);
void rdsp_record_queue_freeBuffer(rdsp_node_t *q);
/* AudioPlayQueue, CONV:344-349 */
rdsp_node_t *rdsp_play_queue_create(rdsp_graph_t *g);
int16_t *rdsp_play_queue_getBuffer(rdsp_node_t *q);
int rdsp_play_queue_playBuffer(rdsp_node_t *q);
/* AudioInputI2S role (INO:52): port 0 = I tile, port 1 = Q tile of this tick */
rdsp_node_t *rdsp_input_node_create(rdsp_graph_t *g);
int rdsp_input_node_push(rdsp_node_t *n, const int16_t *i_tile, const int16_t *q_tile);
/* AudioSDR engine node (INO:53-54,81-86): inputs I,Q; outputs L,R; runs `chain` */
rdsp_node_t *rdsp_sdr_node_create(rdsp_graph_t *g, rdsp_chain_t *chain);
int rdsp_sdr_node_status(rdsp_node_t *n);
int rdsp_chain_decim(const rdsp_chain_t *c);
int rdsp_chain_device(const rdsp_chain_t *c);   /* the device index given to rdsp_chain_create */

/* ---- F1: IQ panadapter spectrum analyser (AudioAnalyzeFFT256IQ, FFTIQ.h:52-110) ----
 * Integer q15 path batched over channels; bit-exact against the oracle.
 *
 * Tables.  Teensy Audio's windows.c / sqrt_integer.c and CMSIS-DSP are not in the reference tree, but
 * the tables the sketch links are in its shipped firmware image (pre_compiled/RadioDSP_SDR_RX.ino.hex)
 * and the generators below reproduce them entry for entry: AudioWindowHanning256 (INO:144),
 * AudioWindowHanning1024 (INO:147), AudioWindowBlackmanNuttall256 (the constructor's default,
 * FFTIQ.h:56) = min(32767, round(32768 w(i / (N - 1)))); twiddleCoef_4096_q15; the 33-entry guess
 * table of sqrt_uint32_approx (tests/test_firmware_tables.py, tests/golden/firmware_tables.npz).
 * The other window ids follow the same rule from the textbook definitions of the names in
 * FFTIQ.h:30-50; the image does not hold them (not pinned).  A caller with its own table uses the
 * reference's signature, rdsp_spectrum_windowFunction_table. */
enum {
  RDSP_WINDOW_NONE = 0,             /* windowFunction(NULL): FFTIQ.cpp:81 skips the multiply */
  RDSP_WINDOW_HANNING = 1,          /* AudioWindowHanning256 / 1024 */
  RDSP_WINDOW_BLACKMAN_HARRIS = 2,
  RDSP_WINDOW_BLACKMAN_NUTTALL = 3, /* AudioWindowBlackmanNuttall256 */
  RDSP_WINDOW_BARTLETT = 4,
  RDSP_WINDOW_BLACKMAN = 5,
  RDSP_WINDOW_FLATTOP = 6,
  RDSP_WINDOW_NUTTALL = 7,
  RDSP_WINDOW_WELCH = 8,
  RDSP_WINDOW_HAMMING = 9,
  RDSP_WINDOW_COSINE = 10,
  RDSP_WINDOW_TUKEY = 11
};
typedef struct rdsp_spectrum rdsp_spectrum_t;
void rdsp_window_q15(int window_id, int16_t *w256);
void rdsp_window_q15_n(int window_id, int n, int16_t *w);
uint32_t rdsp_sqrt_uint32_approx(uint32_t in); /* FFTIQ.cpp:105 (Teensy utility/sqrt_integer.h), host twin of the device routine */
int rdsp_spectrum_create(int n_channels, int device, int naverage, int window_id, rdsp_spectrum_t **out);
int rdsp_spectrum_create_default(int n_channels, int device, rdsp_spectrum_t **out); /* FFTIQ.h:55-58: BlackmanNuttall256, naverage 8 */
void rdsp_spectrum_destroy(rdsp_spectrum_t *s);
int rdsp_spectrum_device(const rdsp_spectrum_t *s);
int rdsp_spectrum_averageTogether(rdsp_spectrum_t *s, int n);      /* FFTIQ.h:88 */
int rdsp_spectrum_windowFunction(rdsp_spectrum_t *s, int window_id); /* FFTIQ.h:93 by table name */
/* void windowFunction(const int16_t *w), FFTIQ.h:93-95, as the reference declares it: a host pointer
 * to 256 q15 taps (copied), NULL = no window */
int rdsp_spectrum_windowFunction_table(rdsp_spectrum_t *s, const int16_t *w256);
int rdsp_spectrum_outputs_for(const rdsp_spectrum_t *s, int n_blocks);
/* float read(unsigned int binNumber), FFTIQ.h:70-73, on one output row (uint16 [256]) */
float rdsp_spectrum_read(const uint16_t *output256, unsigned int binNumber);
/* float read(unsigned int binFirst, unsigned int binLast), FFTIQ.h:75-86, loop as written (`do ... while
 * (binFirst < binLast)`: bins binFirst .. binLast - 1; the single bin when the two are equal) */
float rdsp_spectrum_read_range(const uint16_t *output256, unsigned int binFirst, unsigned int binLast);
/* n_blocks update() ticks (FFTIQ.cpp:65); d_out uint16 [n_channels][out_stride][256] */
int rdsp_spectrum_update(rdsp_spectrum_t *s, const int16_t *d_iq, size_t in_stride, int n_blocks,
                         uint16_t *d_out, size_t out_stride, int *n_outputs, void *stream);

/* the analyser as a node of the block graph (`AudioAnalyzeFFT256IQ FFT;` wired to the
 * pre-processor's I and Q outputs, INO:57,73-74): 2 inputs, no outputs */
rdsp_node_t *rdsp_spectrum_node_create(rdsp_graph_t *g, rdsp_spectrum_t *spec);
int rdsp_spectrum_node_available(rdsp_node_t *n);            /* FFTIQ.h:62-68 */
const uint16_t *rdsp_spectrum_node_output(rdsp_node_t *n);   /* FFTIQ.h:99, [n_channels][256] */
float rdsp_spectrum_node_read(rdsp_node_t *n, int ch, unsigned int binNumber);                          /* FFTIQ.h:70 */
float rdsp_spectrum_node_read_range(rdsp_node_t *n, int ch, unsigned int binFirst, unsigned int binLast); /* FFTIQ.h:75 */
int rdsp_spectrum_node_status(rdsp_node_t *n);

/* ---- F3: biquad cascades ----------------------------------------------------------------------------------------
 * Two different routines of two different libraries, neither in the reference tree, both restated from their published
 * sources and both confirmed in structure by the code of the reference's firmware image (tests/test_firmware_tables.py):
 *
 * (1) the engine's IIR audio filter bank (CTL:153-177 / SURVEY Appendix C): CMSIS-DSP's arm_biquad_cascade_df1_f32 --
 *     direct form 1 in float, acc = (b0 Xn) + (b1 Xn1) + (b2 Xn2) + (a1 Yn1) + (a2 Yn2), every product rounded
 *     before it is added, feedback coefficients stored negated.  In the chain: rdsp_sdr_setAudioFilterKind below.
 *     Design helpers (host): */
void rdsp_biquad_design(int kind, double freq, double q, double fs, float *coef5); /* 0 LP, 1 HP, 2 BP, 3 notch */
void rdsp_design_audio_iir(double f1, double f2, double fs, float *coef20);        /* 8th-order Butterworth band-pass */
/* (2) `AudioFilterBiquad biquad1, biquad2;` (INO:58-59,75-78; `setHighpass(0, 500, 0.5)` INO:155-156): the Teensy Audio
 *     library's FIXED-POINT cascade -- coefficients int32 x 2^30 (a1, a2 stored negated), five 32 x 16 products per
 *     sample that keep the top 32 of 48 bits (SMLAWB / SMLAWT), accumulated on the 14 fractional bits the previous
 *     sample left (`sum &= 0x3FFF`), output `signed_saturate_rshift(sum, 16, 14)`; a fresh object passes nothing
 *     (all-zero coefficients); update() runs stage 0 and goes on to stage s + 1 only if setCoefficients(s + 1) was
 *     ever called; a setter clears its stage's residue and keeps its sample history.  int16 in, int16 out, bit-exact
 *     against the test restatement.  fs: AUDIO_SAMPLE_RATE_EXACT (44100.0 in the reference's image). */
typedef struct rdsp_biquad rdsp_biquad_t; /* AudioFilterBiquad for n_channels streams */
int rdsp_biquad_create(int n_channels, int device, double fs, rdsp_biquad_t **out);
void rdsp_biquad_destroy(rdsp_biquad_t *b);
/* setCoefficients(stage, const double *{b0, b1, b2, a1, a2}) with H = (b0 + b1/z + b2/z^2) / (1 + a1/z + a2/z^2): each
 * coefficient x 1073741824.0 converted to int; setCoefficients(stage, const int *) takes them already scaled */
int rdsp_biquad_setCoefficients(rdsp_biquad_t *b, int stage, const double *coefficients);
int rdsp_biquad_setCoefficients_int(rdsp_biquad_t *b, int stage, const int32_t *coefficients);
int rdsp_biquad_setLowpass(rdsp_biquad_t *b, int stage, float frequency, float q);
int rdsp_biquad_setHighpass(rdsp_biquad_t *b, int stage, float frequency, float q);  /* INO:155-156 */
int rdsp_biquad_setBandpass(rdsp_biquad_t *b, int stage, float frequency, float q);
int rdsp_biquad_setNotch(rdsp_biquad_t *b, int stage, float frequency, float q);
/* what the four setters compute (host, no device): RBJ cookbook in double with w0 = frequency * (2 * 3.141592654 / fs),
 * each coefficient x 2^30 / (1 + alpha) converted to int; a1, a2 as the transfer function writes
 * them (setCoefficients negates).  kind 0 LP, 1 HP, 2 BP, 3 notch */
void rdsp_teensy_biquad_design(int kind, float frequency, float q, float fs, int32_t *coef5);
int rdsp_biquad_get_definition(const rdsp_biquad_t *b, int32_t *out20, int *n_stages); /* b0, b1, b2, -a1, -a2 x 2^30 per stage */
int rdsp_biquad_get_coeffs(const rdsp_biquad_t *b, float *out20);                    /* the same / 2^30 */
/* n_blocks update() ticks: int16 [n_channels][stride] samples read / written every `step` int16
 * (1: planar mono blocks; 2: one side of interleaved pairs, e.g. the I or the Q of an IQ stream) */
int rdsp_biquad_update(rdsp_biquad_t *b, const int16_t *d_in, size_t in_stride, int in_step, int n_blocks,
                       int16_t *d_out, size_t out_stride, int out_step, void *stream);
rdsp_node_t *rdsp_biquad_node_create(rdsp_graph_t *g, rdsp_biquad_t *b); /* 1 input, 1 output */
int rdsp_biquad_node_status(rdsp_node_t *n);
/* Implementation of the engine's audio filter (SDR.enableAudioFilter / setAudioFilter):
 * RDSP_AUDIO_KIND_MASK (default): the pass band is the overlap-save mask (CONV:209-224).
 * RDSP_AUDIO_KIND_IIR: the mask keeps only the side-band selection (50 Hz ... 4 kHz on the
 * demodulator's side) and the selected audio filter is an 8th-order band-pass of four biquads on
 * the demodulated audio, between the overlap-save filter and the NR / notch / AGC stages. */
enum { RDSP_AUDIO_KIND_MASK = 0, RDSP_AUDIO_KIND_IIR = 1 };
int rdsp_sdr_setAudioFilterKind(rdsp_chain_t *c, int kind, void *stream);
int rdsp_chain_get_iir_coeffs(rdsp_chain_t *c, int group, float *out20);
/* An explicit cascade instead of the designed one: coef20 = four sections {b0, b1, b2, a1, a2} in
 * arm_biquad_cascade_df1_f32 order (feedback terms added), e.g. one of the fifteen sets of the engine's own audio
 * filters that the reference's firmware image holds (tests/golden/firmware_tables.npz `biquad_sets`; the first eight
 * are 150 Hz ... 2.1 / 2.3 / 2.5 / 2.7 / 2.9 / 3.1 / 3.3 / 3.9 kHz band-passes for fs = 44 117.647 Hz, each a 4th-order
 * elliptic-type high-pass and low-pass pair: zeros on the unit circle at 21 / 50 Hz and at 2.8 fu / 5.4 fu).  Needs
 * RDSP_AUDIO_KIND_IIR; in force until the next setAudioFilter / setDemodMode of the group -- or until
 * rdsp_chain_set_groups, which re-designs every group's filter from its settings: load explicit sets after regrouping. */
int rdsp_group_setAudioIIRCoefficients(rdsp_chain_t *c, int group, const float *coef20);
int rdsp_sdr_setAudioIIRCoefficients(rdsp_chain_t *c, const float *coef20);

/* ---- AudioAnalyzeFFT1024 (Teensy Audio library; `AudioAnalyzeFFT1024 AudioFFT` on Q_out_L,
 * INO:57,87): 1024-point frames of the audio stream with hop 512 (blocks collected eight at a time,
 * four kept), q15 window, arm_cfft_radix4_q15 of the real samples, output[i] = sqrt_uint32_approx(|X_i|^2)
 * for the 512 bins from DC up (integer, bit-exact against the test restatement; FFT, window, twiddles
 * and square root as for F1).  window_id: RDSP_WINDOW_*. */
typedef struct rdsp_fft1024 rdsp_fft1024_t;
int rdsp_fft1024_create(int n_channels, int device, int window_id, rdsp_fft1024_t **out);
void rdsp_fft1024_destroy(rdsp_fft1024_t *s);
int rdsp_fft1024_windowFunction(rdsp_fft1024_t *s, int window_id); /* INO:147 by table name */
int rdsp_fft1024_windowFunction_table(rdsp_fft1024_t *s, const int16_t *w1024); /* windowFunction(const int16_t *), NULL = none */
float rdsp_fft1024_read(const uint16_t *output512, unsigned int binNumber);   /* AudioAnalyzeFFT1024::read(bin) */
float rdsp_fft1024_read_range(const uint16_t *output512, unsigned int binFirst, unsigned int binLast); /* ::read(first, last), inclusive */
int rdsp_fft1024_averageTogether(rdsp_fft1024_t *s, int n);       /* INO:148: a no-op in the library too */
int rdsp_fft1024_outputs_for(const rdsp_fft1024_t *s, int n_blocks);
/* n_blocks update() ticks; d_audio int16 [n_channels][in_stride] samples taken every in_step int16
 * (2 with an offset pointer picks L or R of the chain's interleaved output); d_out uint16
 * [n_channels][out_stride][512] receives the spectra completed in this call */
int rdsp_fft1024_update(rdsp_fft1024_t *s, const int16_t *d_audio, size_t in_stride, int in_step, int n_blocks,
                        uint16_t *d_out, size_t out_stride, int *n_outputs, void *stream);
rdsp_node_t *rdsp_fft1024_node_create(rdsp_graph_t *g, rdsp_fft1024_t *s); /* 1 input, no outputs */
int rdsp_fft1024_node_available(rdsp_node_t *n);
const uint16_t *rdsp_fft1024_node_output(rdsp_node_t *n); /* [n_channels][512] */
float rdsp_fft1024_node_read(rdsp_node_t *n, int ch, unsigned int binNumber);
float rdsp_fft1024_node_read_range(rdsp_node_t *n, int ch, unsigned int binFirst, unsigned int binLast);
int rdsp_fft1024_node_status(rdsp_node_t *n);

/* ---- deterministic synthetic IQ generator (host) -------------------*/
typedef struct {
  double fs;        /* 96000 */
  double f_off;     /* IF offset, 12000 */
  int32_t cw;       /* 0: two USB tones + interferer; 1: keyed CW tone (K4) */
  double amp_tone;  /* 0.20 */
  double amp_carrier; /* 0.30 */
  double sigma;     /* 0.05 per rail */
} rdsp_synth_config_t;
/* dst: int16 [n_ch][n_samples][2]; channels ch0..ch0+n_ch-1, samples t0..t0+n_samples-1 */
void rdsp_synth_iq(int16_t *dst, int ch0, int n_ch, uint64_t t0, int n_samples,
                   const rdsp_synth_config_t *cfg, int n_threads);

/* ---- `AudioSDR SDR;` as the reference's engine computes it (INO:54; wired INO:81-86) ------------------------------
 * The engine (Derek Rowell's AudioSDR library) is not in the reference tree; its compiled code is, in
 * pre_compiled/RadioDSP_SDR_RX.ino.hex.  rdsp_engine_t follows that code (AudioSDR::update, ITCM 0xe730, and the setters
 * the sketch calls) for n_channels receivers: on the same int16 IQ blocks it returns the int16 audio the image's update()
 * returns (tests/test_engine_kat.py; csrc/rdsp_engine.hip describes the signal path and the kernels).  Native rate only:
 * 44.1 kHz, one 128-sample block in, one out, no decimation.  The rdsp_sdr_* setters above belong to rdsp_chain_t, this
 * build's own many-channel receiver with a decimator in front; these are the reference's.  Numbers are the engine's
 * (`mode`: 0 LSBmode, 1 USBmode, 2 CW_LSBmode, 3 CW_USBmode, 4 AMmode, 5 SAMmode, as the compiled tuningMode() passes
 * them, CTL:337-407; audio filter ids as the compiled filterMode() passes them: 0 audioAM, 1 audioCW, 3 audio2100,
 * 6 audio2700, 8 audio3100, 10 = none, CTL:153-177; AGC modes 0 off ... 3 slow, CTL:200-218). */
typedef struct rdsp_engine rdsp_engine_t;
int rdsp_engine_create(int n_channels, int device, int max_blocks_per_call, rdsp_engine_t **out); /* AudioSDR::AudioSDR */
void rdsp_engine_destroy(rdsp_engine_t *e);
int rdsp_engine_reset(rdsp_engine_t *e, void *stream); /* signal state as the constructor leaves it; settings kept */
/* The engine's tables without a closed form, in the image's order: fifteen sets of four {b0, b1, b2, a1, a2} sections
 * (arm_biquad_cascade_df1_f32 layout) and the 64 taps of one side of its Hilbert transformer, outermost first.
 * rdsp_engine_update fails with RDSP_ERR_NOT_READY until they are loaded. */
int rdsp_engine_load_tables(rdsp_engine_t *e, const float *biquad_sets15x20, const float *hilbert64);
int rdsp_engine_enableAGC(rdsp_engine_t *e);                       /* INO:120 */
int rdsp_engine_setAGCmode(rdsp_engine_t *e, int mode);            /* INO:121, CTL:200-218 */
int rdsp_engine_enableALSfilter(rdsp_engine_t *e);                 /* CTL:259 (clears the filter) */
int rdsp_engine_disableALSfilter(rdsp_engine_t *e);                /* INO:125 */
int rdsp_engine_setALSfilterNotch(rdsp_engine_t *e);               /* CTL:260 */
int rdsp_engine_setALSfilterPeak(rdsp_engine_t *e);                /* backup/RadioDSP_SDR_RX_Conv.ino:665 */
int rdsp_engine_setALSfilterAdaptive(rdsp_engine_t *e);            /* CTL:261 */
int rdsp_engine_enableNoiseBlanker(rdsp_engine_t *e);              /* backup/RadioDSP_SDR_RX_Conv.ino:1259; the constructor's default */
int rdsp_engine_disableNoiseBlanker(rdsp_engine_t *e);             /* INO:131 */
int rdsp_engine_setInputGain(rdsp_engine_t *e, float g);           /* INO:133 */
int rdsp_engine_setOutputGain(rdsp_engine_t *e, float g);          /* INO:134 */
int rdsp_engine_setIQgainBalance(rdsp_engine_t *e, float b);       /* INO:135 */
int rdsp_engine_enableAudioFilter(rdsp_engine_t *e);               /* INO:137 */
int rdsp_engine_setAudioFilter(rdsp_engine_t *e, int id);          /* INO:138, CTL:153-177 */
float rdsp_engine_setDemodMode(rdsp_engine_t *e, int mode);        /* INO:139: returns TuningOffset, Hz */
int rdsp_engine_setMute(rdsp_engine_t *e, int on);                 /* INO:177 */
/* AudioSDR::update() for n_blocks consecutive blocks of every channel, stream-ordered.  d_iq: [ch][t] int16 pairs (I, Q),
 * in_stride pairs between channel rows; d_lr: [ch][t] int16 pairs = the engine's two outputs (the same block on both). */
int rdsp_engine_update(rdsp_engine_t *e, const int16_t *d_iq, size_t in_stride, int n_blocks, int16_t *d_lr,
                       size_t out_stride, void *stream);
/* Receiver groups.  The sketch has one receiver -- one mode, one audio filter, one AGC setting; an object of many
 * channels can be cut into groups of CONSECUTIVE channels that each carry their own settings.  first_channel[g] is group
 * g's first channel (ascending, first_channel[0] = 0; a new group starts as a copy of the group its first channel was in).
 * The setters above address the group chosen with rdsp_engine_select_group (-1, the default: every group);
 * rdsp_engine_setDemodMode returns the offset of the selected group (of group 0 for -1).  A channel's signal state does
 * not care which group it is in: regrouping in mid-stream only changes which settings reach it. */
int rdsp_engine_set_groups(rdsp_engine_t *e, int n_groups, const int *first_channel);
int rdsp_engine_groups(const rdsp_engine_t *e);
int rdsp_engine_select_group(rdsp_engine_t *e, int group);
/* The signal state of a channel range as data (resume; receivers moved between objects or GPUs): filter states, oscillator
 * and detector scalars, AGC, the last 512 samples of the side-band network's lines in time order (independent of either
 * object's ring size), blanker and ALS lines and taps.  Settings are not part of it.  A loaded range continues bit for bit. */
size_t rdsp_engine_state_bytes(const rdsp_engine_t *e, int n_channels);
int rdsp_engine_save_state(rdsp_engine_t *e, int first_channel, int n_channels, void *host_buf, size_t bytes, void *stream);
int rdsp_engine_load_state(rdsp_engine_t *e, int first_channel, const void *host_buf, size_t bytes, void *stream);
int rdsp_engine_channels(const rdsp_engine_t *e);
int rdsp_engine_device(const rdsp_engine_t *e);
int rdsp_engine_max_blocks(const rdsp_engine_t *e);
/* [n_channels][8]: oscillator phase, AGC gain, AGC envelope, hang counter, AGC-active flag, PLL frequency estimate (Hz),
 * PLL lock flag, blanker-hit flag */
int rdsp_engine_get_scalars(rdsp_engine_t *e, float *host_out, void *stream);
const float *rdsp_engine_agc_curve(const rdsp_engine_t *e); /* host copy, 130 entries (the engine uses 129) */
const float *rdsp_engine_sine_table(const rdsp_engine_t *e); /* host copy, 257 entries */

/* ---- `AudioSDRpreProcessor preProcessor;` (INO:53; wired INO:71-72) as the reference's engine library computes it -------
 * (image ::update 0xee88): finds and repairs a one-sample slip between the rails with a 128-point FFT per block while
 * detection is on, and swaps the rails on request.  The chain's own rdsp_pre_* calls above are a different mechanism
 * (a slip set by the host, an estimator on a recording); this object is the reference's. */
typedef struct rdsp_preproc rdsp_preproc_t;
int rdsp_preproc_create(int n_channels, int device, rdsp_preproc_t **out);
void rdsp_preproc_destroy(rdsp_preproc_t *p);
int rdsp_preproc_startAutoI2SerrorDetection(rdsp_preproc_t *p); /* INO:117; takes effect at the next update */
int rdsp_preproc_swapIQ(rdsp_preproc_t *p, int on);             /* INO:118 */
int rdsp_preproc_update(rdsp_preproc_t *p, const int16_t *d_iq, size_t in_stride, int n_blocks, int16_t *d_out,
                        size_t out_stride, void *stream);      /* [ch][t] int16 pairs (I, Q) in and out; in place allowed */
int rdsp_preproc_get_state(rdsp_preproc_t *p, int16_t *host_out, void *stream); /* [ch][4]: remedy (0, 1 = I later, -1 = Q later), bad count, counted blocks, detecting */
int rdsp_preproc_channels(const rdsp_preproc_t *p);
int rdsp_preproc_device(const rdsp_preproc_t *p);

/* the two objects as nodes of the block graph (two inputs I / Q, two outputs, one block per tick): INO:71-72, :81-86 */
rdsp_node_t *rdsp_preproc_node_create(rdsp_graph_t *g, rdsp_preproc_t *p);
rdsp_node_t *rdsp_engine_node_create(rdsp_graph_t *g, rdsp_engine_t *e);
int rdsp_engine_node_status(rdsp_node_t *n); /* either kind */

/* The sketch as shipped inside ONE chain: a chain created as the bare CONV stage (decim 1, 44.1 kHz, RDSP_DEMOD_IQ, no
 * mixer offset, unit gains, AGC / ALS / spectral stage off -- what loop() runs, INO:198) takes the reference's own
 * pre-processor and engine in front of it.  rdsp_chain_process then is INO:71-86 + :198 (preProcessor -> SDR ->
 * doConvolutionalProcessing), rdsp_sdr_node_create wires exactly that into the graph, and the rdsp_sdr_* / rdsp_pre_*
 * setters above reach rdsp_engine_t / rdsp_preproc_t (the image's arithmetic) instead of this build's stand-ins; mode and
 * filter arguments stay rdsp_demod_t / rdsp_audio_filter_t and are translated to the engine's numbers. */
int rdsp_sdr_set_engine_literal(rdsp_chain_t *c, int on);
int rdsp_sdr_load_engine_tables(rdsp_chain_t *c, const float *biquad_sets15x20, const float *hilbert64);
rdsp_engine_t *rdsp_chain_engine(rdsp_chain_t *c);   /* the objects themselves (NULL unless engine-literal) */
rdsp_preproc_t *rdsp_chain_preproc(rdsp_chain_t *c);

#ifdef __cplusplus
}
#endif
#endif
