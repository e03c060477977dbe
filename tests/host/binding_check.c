/*
 * binding_check.c -- a C caller of the boundary: setup() and loop() of RadioDSP_SDR_RX.ino
 * (:117-139,172-183,195-198) over tests/host/rdsp_binding.h, with the record/play queues
 * replaced by two raw int16 files.  BASELINE config K1: 1 channel, 96 kHz IQ, 128-sample
 * blocks, NR/notch off.
 *
 *   binding_check <iq_in.raw> <audio_out.raw> <n_blocks> [spectra_out.raw]
 *
 * With the fourth argument the panadapter side of the graph runs too (biquad1 / biquad2 high-pass 500 Hz -> FFT with
 * AudioWindowHanning256 handed over by pointer, averageTogether(30); .ino:57-59,75-78,144-145,155-156) and every
 * spectrum FFT.available() announces is appended to the file, followed by FFT.read(80) and FFT.read(75, 85).
 *
 * Built by tests/test_boundary_c.py with
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include binding_check.c \
 *       -L radiodsp_sdr_rx_amd -lrdsp_hip -L /opt/rocm/lib -lamdhip64
 */
#include <stdlib.h>
#include <string.h>

#define N_CH 1
#define MAX_BLOCKS 16
#include "rdsp_binding.h"

static FILE *f_in, *f_out, *f_spec;
static long blocks_left;
static int16_t h_iq[MAX_BLOCKS * RDSP_BLOCK_SAMPLES * 2], h_out[MAX_BLOCKS * RDSP_BLOCK_SAMPLES / OUT_DIV * 2];

int queued_blocks(void) { return blocks_left > MAX_BLOCKS ? MAX_BLOCKS : (int)blocks_left; }
void upload_queued_iq(int16_t *d_iq, int n_blocks, hipStream_t s) {
  const size_t n = (size_t)n_blocks * RDSP_BLOCK_SAMPLES * 2;
  if (fread(h_iq, sizeof(int16_t), n, f_in) != n) { fprintf(stderr, "short read\n"); exit(2); }
  blocks_left -= n_blocks;
  if (hipMemcpyAsync(d_iq, h_iq, n * sizeof(int16_t), hipMemcpyHostToDevice, s) != hipSuccess) exit(3);
  if (f_spec) { /* the audio ISR ticks the panadapter nodes on the same blocks (.ino:75-78) */
    panadapter_update(d_iq, n_blocks);
    if (FFT_available()) { /* what the display does with it: output[] and read() */
      const float r[2] = {FFT_read(80), FFT_read2(75, 85)};
      fwrite(FFT_output, sizeof(uint16_t), 256, f_spec);
      fwrite(r, sizeof(float), 2, f_spec);
    }
  }
}
void play_audio(const int16_t *d_out, int n_pairs, hipStream_t s) {
  if (hipMemcpyAsync(h_out, d_out, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, s) != hipSuccess) exit(3);
  if (hipStreamSynchronize(s) != hipSuccess) exit(3);
  fwrite(h_out, 4, (size_t)n_pairs, f_out);
}

static int nr_level = 0;            /* RDSP_general_includes.h:111 */
static uint32_t TuningOffset;

static void setup(void) {           /* RadioDSP_SDR_RX.ino:102-187, the DSP part */
  doConvolutionalInitialize();      /* the chain must exist before the engine setters can reach it */
#if defined(RDSP_BIND_LITERAL) && !defined(RDSP_BIND_ENGINE) /* the CONV stage alone, fed what the image's record queues were fed: no engine in front */
  if (getenv("RDSP_NR_LEVEL")) nr_level = atoi(getenv("RDSP_NR_LEVEL"));
  Init_LMS_NR(15);                            /* :172 */
  reInitializeFilter(300.0, 4000.0);          /* :183 */
  (void)TuningOffset; (void)SDR_setDemodMode;
  return;
#endif
#ifdef RDSP_BIND_ENGINE             /* the sketch as shipped: INO:53-54's objects are the reference's own */
  {
    static float tables[364];
    FILE *ft = getenv("RDSP_ENGINE_TABLES") ? fopen(getenv("RDSP_ENGINE_TABLES"), "rb") : NULL;
    if (!ft || fread(tables, sizeof(float), 364, ft) != 364) { fprintf(stderr, "RDSP_ENGINE_TABLES: 364 floats expected\n"); exit(2); }
    fclose(ft);
    engine_begin(tables);
    nr_level = 15;                            /* RDSP_general_includes.h:111 as the image has it at start-up */
  }
#endif
  preProcessor_startAutoI2SerrorDetection();  /* :117 */
  SDR_enableAGC();                            /* :120 */
  SDR_setAGCmode(AGCmedium);                  /* :121 */
  SDR_disableALSfilter();                     /* :125 */
  SDR_disableNoiseBlanker();                  /* :131 */
  SDR_setInputGain(1.0f);                     /* :133 */
  SDR_setOutputGain(0.5f);                    /* :134 */
  SDR_setIQgainBalance(1.020f);               /* :135 */
  SDR_enableAudioFilter();                    /* :137 */
  SDR_setAudioFilter(audio2700);              /* :138 */
  TuningOffset = SDR_setDemodMode(LSBmode);   /* :139 */
  if (f_spec) {
    panadapter_begin();                       /* :57-59 */
    FFT_windowFunction(AudioWindowHanning256);/* :144 */
    FFT_averageTogether(30);                  /* :145 */
    biquad1_setHighpass(0, 500, 0.5f);        /* :155 */
    biquad2_setHighpass(0, 500, 0.5f);        /* :156 */
  }
  Init_LMS_NR(15);                            /* :172 */
  SDR_setMute(0);                             /* :177 */
  reInitializeFilter(300.0, 4000.0);          /* :183 */
}

static void loop(void) { doConvolutionalProcessing((float)nr_level, 1, 300.0, 4000.0); } /* :195-198 */

int main(int argc, char **argv) {
  if (argc != 4 && argc != 5) { fprintf(stderr, "usage: %s iq_in.raw audio_out.raw n_blocks [spectra_out.raw]\n", argv[0]); return 64; }
  f_in = fopen(argv[1], "rb");
  f_out = fopen(argv[2], "wb");
  if (argc == 5 && !(f_spec = fopen(argv[4], "wb"))) { fprintf(stderr, "cannot open %s\n", argv[4]); return 66; }
  blocks_left = atol(argv[3]);
  if (!f_in || !f_out || blocks_left <= 0) { fprintf(stderr, "cannot open files\n"); return 66; }
  setup();
  while (g_binding_status == RDSP_OK && queued_blocks() >= rdsp_chain_granule_blocks(g_chain)) loop();
  fclose(f_out);
  fclose(f_in);
  if (f_spec) fclose(f_spec);
  (void)FFT_read; (void)FFT_read2; (void)FFT_available; (void)panadapter_update; (void)panadapter_begin;
  if (g_chain) rdsp_chain_destroy(g_chain);
  printf("binding_check: status %d, TuningOffset %u, %s\n", g_binding_status, (unsigned)TuningOffset, rdsp_version());
  return g_binding_status == RDSP_OK ? 0 : 1;
}
