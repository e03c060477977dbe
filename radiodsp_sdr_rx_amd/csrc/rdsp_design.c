/*
 * rdsp_design.c -- host-side (plain C) setup math of the receive chain:
 * filter design, filter mask, decimator taps, NCO tables, synthetic IQ.
 * Runs once per retune, never on the streaming path (the reference does it
 * under AudioNoInterrupts, RDSP_convolutional.h:209-224).
 */
#include "rdsp_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const double kPi = 3.14159265358979323846;

/* cosine-sum window families of RDSP_convolutional.h:152-179 as coefficient rows
 * a0 - a1 cos(t) + a2 cos(2t) - a3 cos(3t), t = 2*pi*i/(n-1) */
static double cosine_sum_window(int id, int i, int n) {
  static const double rows[3][4] = {
      {0.35875, 0.48829, 0.14128, 0.01168},       /* 1: 4-term Blackman-Harris */
      {0.355768, 0.487396, 0.144232, 0.012604},   /* 2 */
      {0.3635819, 0.4891775, 0.1365995, 0.0106411} /* default: Blackman-Nuttall */
  };
  const double t = 2.0 * kPi * (double)i / (double)(n - 1);
  if (id == 3) return cos(0.5 * t);          /* cosine: cos(pi*i/(n-1)) */
  if (id == 4) return 0.5 * (1.0 - cos(t));  /* Hann */
  const double *a = rows[id == 1 ? 0 : (id == 2 ? 1 : 2)];
  return a[0] - a[1] * cos(t) + a[2] * cos(2.0 * t) - a[3] * cos(3.0 * t);
}

/* RDSP_convolutional.h:127-185: windowed-sinc low-pass of half-width
 * (FHi-FLo)/2, complex-shifted to (FHi+FLo)/2. */
void rdsp_calc_cplx_FIR_coeffs(double *coeffs_I, double *coeffs_Q, int numCoeffs, double FLoCut,
                               double FHiCut, double SampleRate, int window) {
  const double lo = FLoCut / SampleRate, hi = FHiCut / SampleRate;
  const double half_bw = 0.5 * (hi - lo);
  const double shift = kPi * (hi + lo);
  const double mid = 0.5 * (double)(numCoeffs - 1);
  for (int i = 0; i < numCoeffs; i++) {
    const double x = (double)i - mid;
    double proto;
    if (fabs(x) < 0.01)
      proto = 2.0 * half_bw; /* odd-length centre tap, :149-150 */
    else
      proto = sin(2.0 * kPi * x * half_bw) / (kPi * x) * cosine_sum_window(window, i, numCoeffs);
    coeffs_I[i] = proto * cos(shift * x);
    coeffs_Q[i] = proto * sin(shift * x);
  }
}

/* time-domain image that RDSP_convolutional.h:96-105 feeds to the FFT: taps
 * narrowed to float, zero padded; the zero fill starts at float index
 * FFT_length+1, which clears the imaginary part of the last tap. */
static void mask_time_image(double *re, double *im, const double *coef_I, const double *coef_Q,
                            int n) {
  const int ntaps = n / 2 + 1;
  for (int i = 0; i < n; i++) re[i] = im[i] = 0.0;
  for (int i = 0; i < ntaps; i++) {
    re[i] = (double)(float)coef_I[i];
    im[i] = (double)(float)coef_Q[i];
  }
  im[n / 2] = 0.0;
}

/* double-precision transform of the tap image: iterative radix-2, decimation in time, n a power of
 * two (256 ... 4096).  The reference feeds the float taps to arm_cfft_f32 (CONV:106); the mask is a
 * design constant, so it is evaluated in double and narrowed once -- and in n log n, because a
 * retune is a host call on the control path of every receiver group: the direct
 * sum this replaces took 0.3 ms at n = 256 and 5 ... 20 ms at n = 4096. */
void rdsp_host_fft(double *re, double *im, int n) {
  for (int i = 1, j = 0; i < n; i++) { /* bit reversal */
    int bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  double *cs = (double *)malloc(sizeof(double) * (size_t)n); /* cos, -sin of 2 pi k / n, k < n / 2 */
  for (int k = 0; k < n / 2; k++) {
    cs[2 * k] = cos(2.0 * kPi * (double)k / (double)n);
    cs[2 * k + 1] = -sin(2.0 * kPi * (double)k / (double)n);
  }
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < half; k++) {
        const double wr = cs[2 * k * step], wi = cs[2 * k * step + 1];
        const double xr = re[i + k + half], xi = im[i + k + half];
        const double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
        re[i + k + half] = re[i + k] - tr;
        im[i + k + half] = im[i + k] - ti;
        re[i + k] += tr;
        im[i + k] += ti;
      }
  }
  free(cs);
}

int rdsp_init_filter_mask(float *mask, const double *coef_I, const double *coef_Q, int fft_l) {
  if (rdsp_plan_radix(fft_l) == 0) return -1;
  const int n = fft_l;
  double *buf = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  if (!buf) return -6;
  double *re = buf, *im = buf + n;
  mask_time_image(re, im, coef_I, coef_Q, n);
  rdsp_host_fft(re, im, n);
  for (int k = 0; k < n; k++) {
    mask[2 * k] = (float)re[k];
    mask[2 * k + 1] = (float)im[k];
  }
  free(buf);
  return 0;
}

/* radix P used by the kernels for each FFT_L (rdsp_kernels.hip dispatch) */
int rdsp_plan_radix(int fft_l) {
  switch (fft_l) {
    case 256: return 4;
    case 512: return 8;
    case 1024: return 16;
    case 2048: return 8;
    case 4096: return 16;
    default: return 0;
  }
}

static int ilog2i(int x) {
  int l = 0;
  while ((1 << l) < x) l++;
  return l;
}

/* natural bin held at position i after the forward transform
 * (mirrors rdsp::bin_of_pos in rdsp_fft.h) */
int rdsp_bin_of_pos(int fft_l, int i) {
  const int P = rdsp_plan_radix(fft_l);
  const int logp = ilog2i(P), logn = ilog2i(fft_l);
  const int nfull = logn / logp;
  const int rem = logn % logp;
  const int np = rem ? nfull + 1 : nfull;
  const int rl = rem ? (1 << rem) : P;
  int k = 0, mult = 1;
  for (int p = 0; p < np; p++) {
    const int s = (p >= np - 1) ? 1 : (fft_l >> (logp * (p + 1)));
    const int R = (p == np - 1) ? rl : P;
    const int digit = (i / s) % R;
    k += digit * mult;
    mult *= R;
  }
  return k;
}

/* device image of the mask: mask/N, digit-reversed, thread-major
 * (element e of thread t at e*NT + t). mask_nat == NULL -> all-pass. */
void rdsp_mask_device_image(const float *mask_nat, int fft_l, float *image) {
  const int P = rdsp_plan_radix(fft_l);
  const int nt = fft_l / P;
  const float inv_n = 1.0f / (float)fft_l;
  for (int t = 0; t < nt; t++)
    for (int e = 0; e < P; e++) {
      const int k = rdsp_bin_of_pos(fft_l, t * P + e);
      const int o = e * nt + t;
      if (mask_nat) {
        image[2 * o] = mask_nat[2 * k] * inv_n;
        image[2 * o + 1] = mask_nat[2 * k + 1] * inv_n;
      } else {
        image[2 * o] = inv_n;
        image[2 * o + 1] = 0.0f;
      }
    }
}

/* decimator low-pass: the same windowed-sinc law with a symmetric band
 * (-B, +B), so the taps are real; reordered by polyphase branch:
 * hc[c][k'] = h[4k' + c] */
int rdsp_design_decimator(int ntaps, double cut_hz, double fs, int window, float *h_nat,
                          float *hc) {
  if (ntaps != 256) return -1;
  double *ti = (double *)malloc(sizeof(double) * 2 * (size_t)ntaps);
  if (!ti) return -6;
  double *tq = ti + ntaps;
  rdsp_calc_cplx_FIR_coeffs(ti, tq, ntaps, -cut_hz, cut_hz, fs, window);
  for (int k = 0; k < ntaps; k++) h_nat[k] = (float)ti[k];
  for (int c = 0; c < 4; c++)
    for (int kp = 0; kp < 64; kp++) hc[c * 64 + kp] = h_nat[4 * kp + c];
  free(ti);
  return 0;
}

/* Spectra of the decimator's polyphase branches for the frequency-domain decimator
 * (rdsp_front_fd_kernel): g_r[k] = h[4k - r], k = 0..64 (zero outside the 256 taps), N-point
 * DFT in double precision, scaled by 1/N, stored like the filter mask (digit-reversed,
 * thread-major: element e of thread t at e*NT + t), branch r at image + 2*r*N floats. */
int rdsp_fd_decimator_image(const float *h_nat, int fft_l, float *image) {
  const int P = rdsp_plan_radix(fft_l);
  if (P == 0 || fft_l < 128) return -1;
  const int nt = fft_l / P;
  const double inv_n = 1.0 / (double)fft_l;
  for (int r = 0; r < 4; r++) {
    double g[65];
    for (int k = 0; k <= 64; k++) {
      const int t = 4 * k - r;
      g[k] = (t >= 0 && t < 256) ? (double)h_nat[t] : 0.0;
    }
    for (int t = 0; t < nt; t++)
      for (int e = 0; e < P; e++) {
        const int bin = rdsp_bin_of_pos(fft_l, t * P + e);
        double re = 0.0, im = 0.0;
        for (int k = 0; k <= 64; k++) {
          const int m = (int)(((long long)bin * k) % fft_l);
          const double a = -2.0 * kPi * (double)m / (double)fft_l;
          re += g[k] * cos(a);
          im += g[k] * sin(a);
        }
        const size_t o = (size_t)r * (size_t)fft_l + (size_t)e * (size_t)nt + (size_t)t;
        image[2 * o] = (float)(re * inv_n);
        image[2 * o + 1] = (float)(im * inv_n);
      }
  }
  return 0;
}

/* The same branch spectra for the decimator on 16-lane rows (rdsp_front_rd_kernel): 256-point windows, 16 points per
 * lane, bin i + 16 e of lane i at [r][16 e + i]; /256 (the inverse transform is unnormalised). */
int rdsp_rd_decimator_image(const float *h_nat, float *image) {
  const int n = 256;
  for (int r = 0; r < 4; r++) {
    double g[65];
    for (int k = 0; k <= 64; k++) {
      const int t = 4 * k - r;
      g[k] = (t >= 0 && t < 256) ? (double)h_nat[t] : 0.0;
    }
    for (int bin = 0; bin < n; bin++) {
      double re = 0.0, im = 0.0;
      for (int k = 0; k <= 64; k++) {
        const int m = (bin * k) % n;
        const double a = -2.0 * kPi * (double)m / (double)n;
        re += g[k] * cos(a);
        im += g[k] * sin(a);
      }
      const size_t o = (size_t)r * n + (size_t)(bin >> 4) * 16 + (size_t)(bin & 15);
      image[2 * o] = (float)(re / n);
      image[2 * o + 1] = (float)(im / n);
    }
  }
  return 0;
}

/* ---- biquad design -------------------------------------------------------
 * AudioFilterBiquad::setLowpass / setHighpass / setBandpass / setNotch (INO:155-156 calls
 * setHighpass(0, 500, 0.5)): the RBJ audio-EQ cookbook sections the Teensy library documents.
 * kind 0 LP, 1 HP, 2 BP (constant peak gain), 3 notch.  Output in the cascade kernel's order
 * {b0, b1, b2, -a1, -a2} (feedback terms are added, as arm_biquad_cascade_df1_f32 stores them). */
void rdsp_biquad_design(int kind, double freq, double q, double fs, float *coef5) {
  const double w0 = 2.0 * kPi * freq / fs;
  const double cs = cos(w0), alpha = sin(w0) / (2.0 * q);
  const double a0 = 1.0 + alpha;
  double num[3];
  if (kind == 0) { num[1] = 1.0 - cs; num[0] = num[2] = 0.5 * num[1]; }
  else if (kind == 1) { num[1] = -(1.0 + cs); num[0] = num[2] = -0.5 * num[1]; }
  else if (kind == 2) { num[0] = alpha; num[1] = 0.0; num[2] = -alpha; }
  else { num[0] = num[2] = 1.0; num[1] = -2.0 * cs; }
  for (int i = 0; i < 3; i++) coef5[i] = (float)(num[i] / a0);
  coef5[3] = (float)(2.0 * cs / a0);
  coef5[4] = (float)(-(1.0 - alpha) / a0);
}

/* The engine's audio filter bank (SDR.setAudioFilter, CTL:153-177): SURVEY Appendix C reads
 * 8th-order band-passes, -3 dB at 150 Hz and at 2.1 ... 3.9 kHz, out of the firmware's
 * coefficient table.  Here: Butterworth, 4th-order low-pass prototype -> band-pass -> bilinear
 * transform with pre-warped edges, four sections each with zeros at z = +1 and z = -1, unit gain
 * at the geometric centre, the gain shared equally by the sections.  Complex arithmetic in
 * double; the float coefficients are what the kernel and the state read-back use. */
#include <complex.h>
/* AudioFilterBiquad::setLowpass / setHighpass / setBandpass / setNotch of the Teensy Audio library (filter_biquad.h) as
 * published: the RBJ cookbook in double, the angle `frequency * (2 * 3.141592654 / AUDIO_SAMPLE_RATE_EXACT)` in DOUBLE too
 * (the reference's setup() holds `setHighpass(0, 500, 0.5)` folded to five integer literals: only the all-double
 * evaluation reproduces them, a float angle is 8 ... 16 counts off; tests/test_firmware_tables.py), every coefficient
 * times 2^30 / (1 + alpha) and converted to int.  coef5 = {b0, b1, b2, a1, a2} as the library hands them to
 * setCoefficients(stage, const int *).  kind 0 LP, 1 HP, 2 BP, 3 notch. */
static int32_t sat_i32(double x) { /* the Cortex-M conversion (VCVT.S32.F64) saturates; a plain cast past the range is undefined here */
  if (!(x == x)) return 0;
  if (x >= 2147483647.0) return INT32_MAX;
  if (x <= -2147483648.0) return INT32_MIN;
  return (int32_t)x;
}
void rdsp_teensy_biquad_design(int kind, float frequency, float q, float fs, int32_t *coef5) {
  const double w0 = (double)frequency * (2.0 * 3.141592654 / (double)fs);
  const double sinW0 = sin(w0);
  const double alpha = sinW0 / ((double)q * 2.0);
  const double cosW0 = cos(w0);
  const double scale = 1073741824.0 / (1.0 + alpha);
  if (kind == 0) {
    coef5[0] = sat_i32(((1.0 - cosW0) / 2.0) * scale);
    coef5[1] = sat_i32((1.0 - cosW0) * scale);
    coef5[2] = coef5[0];
  } else if (kind == 1) {
    coef5[0] = sat_i32(((1.0 + cosW0) / 2.0) * scale);
    coef5[1] = sat_i32(-(1.0 + cosW0) * scale);
    coef5[2] = coef5[0];
  } else if (kind == 2) {
    coef5[0] = sat_i32(alpha * scale);
    coef5[1] = 0;
    coef5[2] = sat_i32((-alpha) * scale);
  } else {
    coef5[0] = sat_i32(scale);
    coef5[1] = sat_i32((-2.0 * cosW0) * scale);
    coef5[2] = coef5[0];
  }
  coef5[3] = sat_i32((-2.0 * cosW0) * scale);
  coef5[4] = sat_i32((1.0 - alpha) * scale);
}

void rdsp_design_audio_iir(double f1, double f2, double fs, float *coef20) {
  const double k = 2.0 * fs;
  const double wa = k * tan(kPi * f1 / fs), wb = k * tan(kPi * f2 / fs);
  const double bw = wb - wa, w0sq = wa * wb;
  double complex zp[4];
  int n = 0;
  for (int m = 0; m < 2; m++) { /* the two prototype poles of the upper half plane */
    const double complex p = cexp(I * kPi * (2.0 * m + 5.0) / 8.0);
    const double complex hb = 0.5 * bw * p, root = csqrt(hb * hb - w0sq);
    const double complex sp[2] = {hb - root, hb + root};
    for (int i = 0; i < 2; i++) {
      double complex sa = sp[i];
      if (cimag(sa) < 0.0) sa = conj(sa);
      zp[n++] = (k + sa) / (k - sa);
    }
  }
  const double complex zc = cexp(-I * 2.0 * kPi * sqrt(f1 * f2) / fs); /* z^-1 at the centre */
  double complex h = 1.0;
  for (int i = 0; i < 4; i++) {
    const double a1 = -2.0 * creal(zp[i]), a2 = creal(zp[i] * conj(zp[i]));
    h *= (1.0 - zc * zc) / (1.0 + a1 * zc + a2 * zc * zc);
  }
  const double g = pow(cabs(h), -0.25);
  for (int i = 0; i < 4; i++) {
    float *c = coef20 + 5 * i;
    c[0] = (float)g;
    c[1] = 0.0f;
    c[2] = (float)-g;
    c[3] = (float)(2.0 * creal(zp[i]));
    c[4] = (float)(-creal(zp[i] * conj(zp[i])));
  }
}

/* NCO: phase increment in turns*2^32; constant rotations for k samples */
/* sinTable_f32 of CMSIS-DSP's arm_sin_f32 / arm_cos_f32 (SPEC:231-232): 513 entries sin(2 pi k / 512), which the
 * published source lists as decimal literals with eight places ("0.01227154f") -- the table is the float nearest
 * to each of THOSE, not to the sine itself (the two differ in the last bit of about one entry in five) */
void rdsp_arm_sin_table(float *tab513) {
  for (int k = 0; k <= 512; k++) {
    const double v = sin(2.0 * kPi * (double)k / 512.0);
    tab513[k] = (float)(floor(fabs(v) * 1e8 + 0.5) / 1e8 * (v < 0.0 ? -1.0 : 1.0));
  }
}

uint32_t rdsp_nco_dphi(double hz, double fs) {
  const double turns = hz / fs;
  const long long q = llround(turns * 4294967296.0);
  return (uint32_t)(unsigned long long)q;
}
void rdsp_nco_rot(uint32_t dphi, int k, float *out2) {
  const uint32_t ph = dphi * (uint32_t)k;
  const double a = 2.0 * kPi * (double)ph / 4294967296.0;
  out2[0] = (float)cos(a);
  out2[1] = (float)-sin(a);
}

/* NR:48-56: mu from the "DSP strength" setting */
float rdsp_lms_mu(int strength) {
  float m = (float)strength;
  m /= 2;
  m += 2;
  m /= 10;
  m = powf(10, m);
  return 1 / m;
}

/* loop constants of the SAM demodulator's PLL (build-defined: zeta 0.65, omegaN 200 rad/s,
 * lock range +-2 kHz) at the decimated rate */
void rdsp_sam_constants(double fs_out, float *g1, float *g2, float *wmin, float *wmax) {
  const double zeta = 0.65, omegaN = 200.0, fmax = 2000.0;
  const double a = 1.0 - exp(-2.0 * omegaN * zeta / fs_out);
  const double b = -a + 2.0 * (1.0 - exp(-omegaN * zeta / fs_out) * cos(omegaN / fs_out * sqrt(1.0 - zeta * zeta)));
  *g1 = (float)a;
  *g2 = (float)b;
  *wmax = (float)(2.0 * kPi * fmax / fs_out);
  *wmin = -*wmax;
}

/* ---- synthetic IQ: counter-based so any (channel, time) window
 * can be generated independently and identically on any host ---------------- */
static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline double u01(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

static inline int16_t quant16(double v) {
  double q = floor(v * 32767.0 + 0.5);
  if (q > 32767.0) q = 32767.0;
  if (q < -32768.0) q = -32768.0;
  return (int16_t)q;
}

void rdsp_synth_iq(int16_t *dst, int ch0, int n_ch, uint64_t t0, int n_samples,
                   const rdsp_synth_config_t *cfg, int n_threads) {
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#else
  (void)n_threads;
#endif
  const double fs = cfg->fs;
  const double f1 = (cfg->f_off + 700.0) / fs, f2 = (cfg->f_off + 1900.0) / fs,
               f3 = (cfg->f_off + 1000.0) / fs;
  const uint64_t key_len = (uint64_t)(0.06 * fs);
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < n_ch; c++) {
    const uint64_t chn = (uint64_t)(ch0 + c);
    const uint64_t seed = 0x5DEECE66Dull ^ (chn * 0x9E3779B97F4A7C15ull);
    const double p1 = 2.0 * kPi * u01(splitmix64(seed ^ 0xA5A5A5A5A5A5A5A5ull));
    const double p2 = 2.0 * kPi * u01(splitmix64(seed ^ 0x5A5A5A5A5A5A5A5Aull));
    const double p3 = 2.0 * kPi * u01(splitmix64(seed ^ 0x3C3C3C3C3C3C3C3Cull));
    int16_t *row = dst + (size_t)c * (size_t)n_samples * 2;
    for (int i = 0; i < n_samples; i++) {
      const uint64_t t = t0 + (uint64_t)i;
      const double td = (double)t;
      /* complex white noise, Box-Muller on two counter-indexed draws */
      const double u1 = u01(splitmix64(seed + 2 * t)) + (1.0 / 9007199254740992.0);
      const double u2 = u01(splitmix64(seed + 2 * t + 1));
      const double r = cfg->sigma * sqrt(-2.0 * log(u1));
      double xi = r * cos(2.0 * kPi * u2), xq = r * sin(2.0 * kPi * u2);
      if (cfg->cw) {
        if (((t / key_len) & 1ull) == 0) {
          const double a = 2.0 * kPi * fmod(f1 * td, 1.0) + p1;
          xi += cfg->amp_tone * cos(a);
          xq += cfg->amp_tone * sin(a);
        }
      } else {
        const double a1 = 2.0 * kPi * fmod(f1 * td, 1.0) + p1;
        const double a2 = 2.0 * kPi * fmod(f2 * td, 1.0) + p2;
        const double a3 = 2.0 * kPi * fmod(f3 * td, 1.0) + p3;
        xi += cfg->amp_tone * (cos(a1) + cos(a2)) + cfg->amp_carrier * cos(a3);
        xq += cfg->amp_tone * (sin(a1) + sin(a2)) + cfg->amp_carrier * sin(a3);
      }
      row[2 * i] = quant16(xi);
      row[2 * i + 1] = quant16(xq);
    }
  }
}
