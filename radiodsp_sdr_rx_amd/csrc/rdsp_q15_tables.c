/*
 * rdsp_q15_tables.c -- the constant tables of the two integer analysers (host, plain C) and the
 * read() accessors of their output rows.
 *
 * The Teensy Audio library (windows.c, utility/sqrt_integer.c) and CMSIS-DSP (twiddleCoef_4096_q15)
 * are not in the reference tree, but the tables the sketch links are in its shipped firmware image
 * (pre_compiled/RadioDSP_SDR_RX.ino.hex).  Every generator here reproduces the table the image
 * holds exactly (tests/test_firmware_tables.py against tests/golden/firmware_tables.npz):
 *   AudioWindowHanning256 / Hanning1024 / BlackmanNuttall256 = min(32767, round(32768 w(i/(N-1))))
 *   twiddleCoef_4096_q15[k] = (clamp(floor(32768 cos(2 pi k/4096))), clamp(floor(32768 sin(...))))
 *   sqrt_integer_guess_table[33] as below.
 * The other windows of analyze_fft256iq.h:30-50 follow the same rule from their textbook
 * definitions; the image does not hold them, so those are not pinned.
 */
#include "rdsp_host.h"

#include <math.h>
#include <stddef.h>

static const double kTwoPi = 6.28318530717958647692;

/* w(x), x = i / (N - 1) in [0, 1] */
static double window_value(int id, double x) {
  const double c1 = cos(kTwoPi * x), c2 = cos(2.0 * kTwoPi * x), c3 = cos(3.0 * kTwoPi * x);
  switch (id) {
    case RDSP_WINDOW_HANNING: return 0.5 * (1.0 - c1);
    case RDSP_WINDOW_BLACKMAN_HARRIS: return 0.35875 - 0.48829 * c1 + 0.14128 * c2 - 0.01168 * c3;
    case RDSP_WINDOW_BLACKMAN_NUTTALL: return 0.3635819 - 0.4891775 * c1 + 0.1365995 * c2 - 0.0106411 * c3;
    case RDSP_WINDOW_BARTLETT: return 1.0 - fabs(2.0 * x - 1.0);
    case RDSP_WINDOW_BLACKMAN: return 0.42 - 0.5 * c1 + 0.08 * c2;
    case RDSP_WINDOW_FLATTOP:
      return 0.21557895 - 0.41663158 * c1 + 0.277263158 * c2 - 0.083578947 * c3 + 0.006947368 * cos(4.0 * kTwoPi * x);
    case RDSP_WINDOW_NUTTALL: return 0.355768 - 0.487396 * c1 + 0.144232 * c2 - 0.012604 * c3;
    case RDSP_WINDOW_WELCH: return 1.0 - (2.0 * x - 1.0) * (2.0 * x - 1.0);
    case RDSP_WINDOW_HAMMING: return 0.54 - 0.46 * c1;
    case RDSP_WINDOW_COSINE: return sin(0.5 * kTwoPi * x);
    case RDSP_WINDOW_TUKEY: { /* alpha = 0.5 */
      if (x < 0.25) return 0.5 * (1.0 - cos(kTwoPi * 2.0 * x));
      if (x > 0.75) return 0.5 * (1.0 - cos(kTwoPi * 2.0 * (1.0 - x)));
      return 1.0;
    }
    default: return 1.0;
  }
}

/* the q15 table the library would hold for window `window_id` and n points */
void rdsp_window_q15_n(int window_id, int n, int16_t *w) {
  if (n < 2) { /* w(i / (n - 1)) has no meaning for a single point: an empty or one-entry table is left at 0 */
    if (n == 1) w[0] = 0;
    return;
  }
  for (int i = 0; i < n; i++) {
    double q = floor(32768.0 * window_value(window_id, (double)i / (double)(n - 1)) + 0.5);
    if (q > 32767.0) q = 32767.0;
    if (q < -32768.0) q = -32768.0;
    w[i] = (int16_t)q;
  }
}
void rdsp_window_q15(int window_id, int16_t *w256) { rdsp_window_q15_n(window_id, 256, w256); }

/* W_n^m, m = 0 .. 3n/4 - 1, as CMSIS keeps it: (cos, sin) of 2 pi m / n scaled by 32768, floored,
 * clamped to int16 (twiddleCoef_4096_q15 read with stride 4096 / n, arm_cfft_radix4_init_q15).
 * out[m] = cos | sin << 16. */
void rdsp_q15_twiddles(int n, uint32_t *out) {
  for (int m = 0; m < 3 * n / 4; m++) {
    /* evaluate on the 4096-point grid the table is made on */
    const double a = kTwoPi * (double)(m * (4096 / n)) / 4096.0;
    double c = floor(32768.0 * cos(a)), s = floor(32768.0 * sin(a));
    if (c > 32767.0) c = 32767.0;
    if (s > 32767.0) s = 32767.0;
    out[m] = ((uint32_t)(int32_t)c & 0xFFFFu) | ((uint32_t)(int32_t)s << 16);
  }
}

/* utility/sqrt_integer.c of the Teensy Audio library: first guess by count of leading zeros */
static const uint16_t kSqrtGuess[33] = {55109, 38968, 27555, 19484, 13778, 9742, 6889, 4871, 3445, 2436, 1723,
                                        1218,  862,   609,   431,   305,   216,  153,  108,  77,   54,   39,
                                        27,    20,    14,    10,    7,     5,    4,    3,    2,    1,    0};
const uint16_t *rdsp_sqrt_guess_table(void) { return kSqrtGuess; }

/* sqrt_uint32_approx (utility/sqrt_integer.h), FFTIQ.cpp:105: the host twin of the device routine
 * in rdsp_q15.h.  in = 0 reads guess 0 and divides by it: UDIV by zero yields 0 on the Cortex-M7. */
uint32_t rdsp_sqrt_uint32_approx(uint32_t in) {
  uint32_t n = kSqrtGuess[in ? __builtin_clz(in) : 32];
  if (n == 0) return 0;
  n = ((in / n) + n) / 2;
  n = ((in / n) + n) / 2;
  return n;
}

/* float read(unsigned int binNumber), FFTIQ.h:70-73 */
float rdsp_spectrum_read(const uint16_t *output, unsigned int binNumber) {
  if (binNumber > 255) return 0.0f;
  return (float)(output[binNumber]) * (float)(1.0 / 16384.0);
}
/* float read(unsigned int binFirst, unsigned int binLast), FFTIQ.h:75-86, with its loop as written:
 * `do { sum += output[binFirst++]; } while (binFirst < binLast);` adds bins binFirst .. binLast - 1,
 * and the single bin binFirst when the two are equal -- binLast itself is never added. */
float rdsp_spectrum_read_range(const uint16_t *output, unsigned int binFirst, unsigned int binLast) {
  if (binFirst > binLast) {
    const unsigned int tmp = binLast;
    binLast = binFirst;
    binFirst = tmp;
  }
  if (binFirst > 255) return 0.0f;
  if (binLast > 255) binLast = 255;
  uint32_t sum = 0;
  do {
    sum += output[binFirst++];
  } while (binFirst < binLast);
  return (float)sum * (float)(1.0 / 16384.0);
}
/* AudioAnalyzeFFT1024::read of the Teensy Audio library (the display's reader of AudioFFT, INO:57):
 * same shape over 512 bins */
float rdsp_fft1024_read(const uint16_t *output, unsigned int binNumber) {
  if (binNumber > 511) return 0.0f;
  return (float)(output[binNumber]) * (float)(1.0 / 16384.0);
}
float rdsp_fft1024_read_range(const uint16_t *output, unsigned int binFirst, unsigned int binLast) {
  if (binFirst > binLast) {
    const unsigned int tmp = binLast;
    binLast = binFirst;
    binFirst = tmp;
  }
  if (binFirst > 511) return 0.0f;
  if (binLast > 511) binLast = 511;
  uint32_t sum = 0;
  do {
    sum += output[binFirst++];
  } while (binFirst <= binLast);
  return (float)sum * (float)(1.0 / 16384.0);
}
