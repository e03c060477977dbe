/*
 * rdsp_binding.h -- the reference-side binding of INTEGRATION.md section 2, as a file that
 * meets a compiler: drop-in replacements for the free functions of RDSP_convolutional.h /
 * RDSP_noise_reduction.h and for the `SDR.` / `preProcessor.` call sites of
 * RadioDSP_SDR_RX.ino:117-139,177, over the C-ABI of include/rdsp.h.  Plain C.
 *
 * A host port of the sketch provides the three transport shims declared below (the roles of
 * the record/play queues, RDSP_convolutional.h:231-244,344-349) and the two sizes.
 */
#ifndef RDSP_BINDING_H
#define RDSP_BINDING_H

#include <hip/hip_runtime_api.h>
#include <stdio.h>

#include "rdsp.h"

#ifndef N_CH
#define N_CH 1          /* receivers on this GPU */
#endif
#ifndef MAX_BLOCKS
#define MAX_BLOCKS 64   /* 128-sample blocks per call and channel */
#endif
#define IN_STRIDE ((size_t)MAX_BLOCKS * RDSP_BLOCK_SAMPLES)
#ifdef RDSP_BIND_ENGINE  /* the whole sketch as shipped: the reference's own pre-processor and engine in front of its CONV stage */
#define RDSP_BIND_LITERAL
#endif
#ifdef RDSP_BIND_LITERAL /* the CONV stage as the shipped sketch runs it, behind the engine: 44.1 kHz, no mixer, no decimator */
#define OUT_DIV 1
#else
#define OUT_DIV 4
#endif
#define OUT_STRIDE (IN_STRIDE / OUT_DIV)

typedef int boolean; /* Arduino */

/* transport shims supplied by the host program */
int queued_blocks(void);                                        /* Q_in_L.available(), CONV:231 */
void upload_queued_iq(int16_t *d_iq, int n_blocks, hipStream_t s); /* readBuffer/freeBuffer, CONV:236-244 */
void play_audio(const int16_t *d_out, int n_pairs, hipStream_t s); /* getBuffer/playBuffer, CONV:344-349 */

static rdsp_chain_t *g_chain;          /* replaces the globals of RDSP_convolutional.h:34-80 */
static hipStream_t g_stream;
static int16_t *g_d_iq, *g_d_out;      /* device staging for N_CH x MAX_BLOCKS blocks */
static int g_binding_status = RDSP_OK; /* the sketch has no error channel: first failure is kept here */

#define RDSP_BIND_CHECK(call)                                                      \
  do {                                                                             \
    int rc_ = (call);                                                              \
    if (rc_ != RDSP_OK && g_binding_status == RDSP_OK) {                           \
      g_binding_status = rc_;                                                      \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, rdsp_last_error());            \
    }                                                                              \
  } while (0)

static void doConvolutionalInitialize(void) { /* RDSP_convolutional.h:187 */
#ifdef RDSP_BIND_LITERAL
  /* CONV:34-72 as the firmware image has them: SAMPLE_RATE 44100, FFT_L 256, N_BLOCKS 1, 129 taps, window 1, 300 ... 4000 Hz;
   * L / R in, L / R out (CONV:314-318) */
  rdsp_chain_config_t cfg = {
      .fs_in = 44100.0, .decim = 1, .nco_hz = 0.0, .fft_l = 256, .window = 1, .flo_hz = 300.0, .fhi_hz = 4000.0,
      .filter_on = 1, .demod = RDSP_DEMOD_IQ, .agc_mode = RDSP_AGC_OFF, .input_gain = 1.0f, .output_gain = 1.0f,
      .iq_balance = 1.0f};
#else
  rdsp_chain_config_t cfg = {
      .fs_in = 96000.0, .decim = 4, .fir_taps = 256, .fir_cut_hz = 10000.0, .nco_hz = 12000.0,
      .fft_l = 256, .window = 1, .flo_hz = 300.0, .fhi_hz = 4000.0, .filter_on = 1,
      .demod = RDSP_DEMOD_USB, .als_strength = 20, .agc_mode = RDSP_AGC_OFF,
      .input_gain = 1.0f, .output_gain = 1.0f, .iq_balance = 1.0f};
#endif
  if (hipStreamCreate(&g_stream) != hipSuccess || hipMalloc((void **)&g_d_iq, N_CH * IN_STRIDE * 4) != hipSuccess ||
      hipMalloc((void **)&g_d_out, N_CH * OUT_STRIDE * 4) != hipSuccess) {
    g_binding_status = RDSP_ERR_HIP;
    return;
  }
  RDSP_BIND_CHECK(rdsp_chain_create(&cfg, N_CH, /*device*/ 0, MAX_BLOCKS, &g_chain)); /* runs CONV:187 and :209 */
}
static void reInitializeFilter(double lo, double hi) { /* RDSP_convolutional.h:209 */
  RDSP_BIND_CHECK(rdsp_reInitializeFilter(g_chain, lo, hi, g_stream));
}
static void Init_LMS_NR(int strength) { /* RDSP_noise_reduction.h:35 */
  RDSP_BIND_CHECK(rdsp_Init_LMS_NR(g_chain, strength, g_stream));
}
#ifdef RDSP_BIND_ENGINE
static rdsp_preproc_t *g_pre; /* `AudioSDRpreProcessor preProcessor;` INO:53 */
static rdsp_engine_t *g_sdr;  /* `AudioSDR SDR;`                       INO:54 */
/* the constructors of INO:53-54; `tables` = the engine's 15 x 20 biquad coefficients and 64 Hilbert taps (no closed form:
 * a host takes them from the AudioSDR library it has, the tests from tests/golden/firmware_tables.npz) */
static void engine_begin(const float *tables364) {
  RDSP_BIND_CHECK(rdsp_preproc_create(N_CH, 0, &g_pre));
  RDSP_BIND_CHECK(rdsp_engine_create(N_CH, 0, MAX_BLOCKS, &g_sdr));
  RDSP_BIND_CHECK(rdsp_engine_load_tables(g_sdr, tables364, tables364 + 300));
}
/* the audio interrupt's part of a tick, n blocks at a time: IQinput -> preProcessor -> SDR -> record queues (INO:71-72,81-86) */
static void engine_update(int16_t *d_iq, int n) {
  RDSP_BIND_CHECK(rdsp_preproc_update(g_pre, d_iq, IN_STRIDE, n, d_iq, IN_STRIDE, g_stream));
  RDSP_BIND_CHECK(rdsp_engine_update(g_sdr, d_iq, IN_STRIDE, n, d_iq, IN_STRIDE, g_stream));
}
#endif
static void doConvolutionalProcessing(float nr, boolean filt, double lo, double hi) { /* CONV:228 */
  int n = queued_blocks();                     /* Q_in_L.available(), CONV:231 */
  if (n > MAX_BLOCKS) n = MAX_BLOCKS;
  n -= n % rdsp_chain_granule_blocks(g_chain);
  if (n <= 0) return;                          /* the same silent skip as the sketch */
  upload_queued_iq(g_d_iq, n, g_stream);
#ifdef RDSP_BIND_ENGINE
  engine_update(g_d_iq, n);                    /* what the record queues hold is the engine's audio */
#endif
  RDSP_BIND_CHECK(rdsp_doConvolutionalProcessing(g_chain, nr, filt, lo, hi, g_d_iq, IN_STRIDE, n, g_d_out,
                                                 OUT_STRIDE, g_stream));
  play_audio(g_d_out, n * RDSP_BLOCK_SAMPLES / OUT_DIV, g_stream);
}

#ifdef RDSP_BIND_ENGINE
/* the engine's own objects behind the sketch's call sites (RadioDSP_SDR_RX.ino:117-139,177), the engine's own numbers */
enum { AGCoff = 0, AGCfast = 1, AGCmedium = 2, AGCslow = 3 };
enum { LSBmode = 0, USBmode = 1, CW_LSBmode = 2, CW_USBmode = 3, AMmode = 4, SAMmode = 5 };
enum { audioAM = 0, audioCW = 1, audio2100 = 3, audio2700 = 6, audio3100 = 8 };
#define preProcessor_startAutoI2SerrorDetection() RDSP_BIND_CHECK(rdsp_preproc_startAutoI2SerrorDetection(g_pre))
#define preProcessor_swapIQ(b)      RDSP_BIND_CHECK(rdsp_preproc_swapIQ(g_pre, (b)))
#define SDR_enableAGC()             RDSP_BIND_CHECK(rdsp_engine_enableAGC(g_sdr))
#define SDR_setAGCmode(m)           RDSP_BIND_CHECK(rdsp_engine_setAGCmode(g_sdr, (m)))
#define SDR_enableALSfilter()       RDSP_BIND_CHECK(rdsp_engine_enableALSfilter(g_sdr))
#define SDR_disableALSfilter()      RDSP_BIND_CHECK(rdsp_engine_disableALSfilter(g_sdr))
#define SDR_setALSfilterNotch()     RDSP_BIND_CHECK(rdsp_engine_setALSfilterNotch(g_sdr))
#define SDR_setALSfilterAdaptive()  RDSP_BIND_CHECK(rdsp_engine_setALSfilterAdaptive(g_sdr))
#define SDR_disableNoiseBlanker()   RDSP_BIND_CHECK(rdsp_engine_disableNoiseBlanker(g_sdr))
#define SDR_setInputGain(g)         RDSP_BIND_CHECK(rdsp_engine_setInputGain(g_sdr, (g)))
#define SDR_setOutputGain(g)        RDSP_BIND_CHECK(rdsp_engine_setOutputGain(g_sdr, (g)))
#define SDR_setIQgainBalance(g)     RDSP_BIND_CHECK(rdsp_engine_setIQgainBalance(g_sdr, (g)))
#define SDR_enableAudioFilter()     RDSP_BIND_CHECK(rdsp_engine_enableAudioFilter(g_sdr))
#define SDR_setAudioFilter(f)       RDSP_BIND_CHECK(rdsp_engine_setAudioFilter(g_sdr, (f)))
#define SDR_setDemodMode(m)         ((uint32_t)rdsp_engine_setDemodMode(g_sdr, (m)))   /* `TuningOffset = SDR.setDemodMode(LSBmode);` INO:139 */
#define SDR_setMute(b)              RDSP_BIND_CHECK(rdsp_engine_setMute(g_sdr, (b)))
#else
/* the engine object keeps its call sites (RadioDSP_SDR_RX.ino:117-139,177) */
enum { AGCoff = RDSP_AGC_OFF, AGCfast = RDSP_AGC_FAST, AGCmedium = RDSP_AGC_MEDIUM, AGCslow = RDSP_AGC_SLOW };
enum { LSBmode = RDSP_DEMOD_LSB, USBmode = RDSP_DEMOD_USB, CW_LSBmode = RDSP_DEMOD_CW_LSB,
       CW_USBmode = RDSP_DEMOD_CW_USB, AMmode = RDSP_DEMOD_AM, SAMmode = RDSP_DEMOD_SAM };
enum { audioCW = RDSP_AUDIO_CW, audio2100 = RDSP_AUDIO_2100, audio2700 = RDSP_AUDIO_2700, audio3100 = RDSP_AUDIO_3100,
       audioAM = RDSP_AUDIO_AM, audioWSPR = RDSP_AUDIO_WSPR };
#define preProcessor_startAutoI2SerrorDetection() RDSP_BIND_CHECK(rdsp_pre_startAutoI2SerrorDetection(g_chain))
#define preProcessor_swapIQ(b)      RDSP_BIND_CHECK(rdsp_pre_swapIQ(g_chain, (b)))
#define preProcessor_setIQslip(s)   RDSP_BIND_CHECK(rdsp_pre_setIQslip(g_chain, (s)))   /* build-defined: correction for recordings */
#define SDR_enableAGC()             RDSP_BIND_CHECK(rdsp_sdr_enableAGC(g_chain))
#define SDR_setAGCmode(m)           RDSP_BIND_CHECK(rdsp_sdr_setAGCmode(g_chain, (m)))
#define SDR_disableALSfilter()      RDSP_BIND_CHECK(rdsp_sdr_disableALSfilter(g_chain))
#define SDR_disableNoiseBlanker()   RDSP_BIND_CHECK(rdsp_sdr_disableNoiseBlanker(g_chain))
#define SDR_setInputGain(g)         RDSP_BIND_CHECK(rdsp_sdr_setInputGain(g_chain, (g)))
#define SDR_setOutputGain(g)        RDSP_BIND_CHECK(rdsp_sdr_setOutputGain(g_chain, (g)))
#define SDR_setIQgainBalance(g)     RDSP_BIND_CHECK(rdsp_sdr_setIQgainBalance(g_chain, (g)))
#define SDR_enableAudioFilter()     RDSP_BIND_CHECK(rdsp_sdr_enableAudioFilter(g_chain))
#define SDR_setAudioFilter(f)       RDSP_BIND_CHECK(rdsp_sdr_setAudioFilter(g_chain, (f), g_stream))
/* TuningOffset = SDR.setDemodMode(LSBmode);   .ino:139 -- the engine answers with where it wants the carrier (8390 Hz
 * for LSBmode ...) and from then on moves it to 0 Hz itself; here the mixer is told */
static inline uint32_t SDR_setDemodMode(int m) {
  const uint32_t off = rdsp_sdr_setDemodMode(g_chain, m, g_stream);
  RDSP_BIND_CHECK(rdsp_sdr_setTuningOffsetHz(g_chain, (double)off));
  return off;
}
#define SDR_setMute(b)              RDSP_BIND_CHECK(rdsp_sdr_setMute(g_chain, (b)))
#endif

/* ---- the panadapter side of the graph: `AudioFilterBiquad biquad1, biquad2; AudioAnalyzeFFT256IQ FFT;`
 * (.ino:57-59), IQinput -> biquad1 / biquad2 -> FFT (.ino:75-78), `biquadN.setHighpass(0, 500, 0.5)` (.ino:155-156),
 * `FFT.windowFunction(AudioWindowHanning256); FFT.averageTogether(30);` (.ino:144-145), read by the display through
 * FFT.available() / FFT.output[] / FFT.read() (analyze_fft256iq.h:62-86,99).  The window table is the caller's, as in
 * the reference (`const int16_t *`): Teensy's windows.c defines it there; a host port fills it once with
 * rdsp_window_q15(RDSP_WINDOW_HANNING, ...), which reproduces the table of the reference's firmware image. */
static rdsp_biquad_t *g_biquad1, *g_biquad2;
static rdsp_spectrum_t *g_fft;
static int16_t AudioWindowHanning256[256];
static int16_t *g_d_filt;              /* device: the two high-passed rails, interleaved like the IQ stream */
static uint16_t *g_d_spec, FFT_output[N_CH * 256]; /* uint16_t output[256] (analyze_fft256iq.h:99), one row per receiver */
static int g_fft_outputflag;

static void panadapter_begin(void) {   /* the constructors of .ino:57-59 */
  rdsp_window_q15(RDSP_WINDOW_HANNING, AudioWindowHanning256);
  RDSP_BIND_CHECK(rdsp_biquad_create(N_CH, 0, 44100.0, &g_biquad1));   /* AUDIO_SAMPLE_RATE_EXACT of the Teensy 4 build */
  RDSP_BIND_CHECK(rdsp_biquad_create(N_CH, 0, 44100.0, &g_biquad2));
  RDSP_BIND_CHECK(rdsp_spectrum_create_default(N_CH, 0, &g_fft));       /* BlackmanNuttall256, naverage 8 */
  if (hipMalloc((void **)&g_d_filt, N_CH * IN_STRIDE * 4) != hipSuccess ||
      hipMalloc((void **)&g_d_spec, sizeof(FFT_output) * MAX_BLOCKS) != hipSuccess)
    g_binding_status = RDSP_ERR_HIP;
}
#define biquad1_setHighpass(st, f, q) RDSP_BIND_CHECK(rdsp_biquad_setHighpass(g_biquad1, (st), (f), (q)))
#define biquad2_setHighpass(st, f, q) RDSP_BIND_CHECK(rdsp_biquad_setHighpass(g_biquad2, (st), (f), (q)))
#define FFT_windowFunction(w)         RDSP_BIND_CHECK(rdsp_spectrum_windowFunction_table(g_fft, (w)))
#define FFT_averageTogether(n)        RDSP_BIND_CHECK(rdsp_spectrum_averageTogether(g_fft, (n)))
/* n_blocks update() ticks of the three nodes on the IQ blocks the chain is about to consume (d_iq as uploaded) */
static void panadapter_update(const int16_t *d_iq, int n_blocks) {
  int n_out = 0;
  RDSP_BIND_CHECK(rdsp_biquad_update(g_biquad1, d_iq, IN_STRIDE, 2, n_blocks, g_d_filt, IN_STRIDE, 2, g_stream));         /* I rail: every 2nd int16 */
  RDSP_BIND_CHECK(rdsp_biquad_update(g_biquad2, d_iq + 1, IN_STRIDE, 2, n_blocks, g_d_filt + 1, IN_STRIDE, 2, g_stream)); /* Q rail */
  RDSP_BIND_CHECK(rdsp_spectrum_update(g_fft, g_d_filt, IN_STRIDE, n_blocks, g_d_spec, MAX_BLOCKS, &n_out, g_stream));
  if (n_out > 0) { /* the display sees the latest completed average */
    for (int c = 0; c < N_CH; c++)
      if (hipMemcpyAsync(FFT_output + 256 * c, g_d_spec + ((size_t)c * MAX_BLOCKS + (size_t)(n_out - 1)) * 256, 512,
                         hipMemcpyDeviceToHost, g_stream) != hipSuccess)
        g_binding_status = RDSP_ERR_HIP;
    if (hipStreamSynchronize(g_stream) != hipSuccess) g_binding_status = RDSP_ERR_HIP;
    g_fft_outputflag = 1;
  }
}
static int FFT_available(void) { const int f = g_fft_outputflag; g_fft_outputflag = 0; return f; } /* analyze_fft256iq.h:62-68 */
static float FFT_read(unsigned int bin) { return rdsp_spectrum_read(FFT_output, bin); }             /* :70-73 */
static float FFT_read2(unsigned int first, unsigned int last) { return rdsp_spectrum_read_range(FFT_output, first, last); } /* :75-86 */

#endif
