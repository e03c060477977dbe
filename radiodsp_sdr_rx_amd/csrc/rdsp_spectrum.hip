/*
 * rdsp_spectrum.hip -- SURVEY 8f row F1: the IQ panadapter spectrum analyser
 * (AudioAnalyzeFFT256IQ, analyze_fft256iq.{h,cpp}) batched over channels.
 * Integer q15 path, bit-exact against the CPU restatement kept with the tests: pack I | Q << 16
 * (FFTIQ.cpp:38-48), q15 window (x*w) >> 15 (:50-63), 256-point fixed-point
 * radix-4 FFT (arm_cfft_radix4_q15 as CMSIS publishes it, :82; rdsp_q15.h), |.|^2 / naverage
 * accumulation (:86-98), sqrt_uint32_approx and the output[255 - (i ^ 128)] reorder (:99-113).
 * Windows, twiddles and the square root's guess table are the ones in the reference's firmware
 * image (rdsp_q15_tables.c).
 *
 * One wave per channel: lane t owns the four inputs t + 64k of a 256-point frame
 * ([previous block | current block], so the previous block simply stays in two
 * registers), runs one radix-4 butterfly per stage and exchanges packed int16
 * pairs through LDS between the four stages.  The interleaved int16 IQ stream is
 * already the packed word the reference builds, so the only HBM traffic is one
 * coalesced 4-byte read per input sample.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rdsp_host.h"
#include "rdsp_q15.h"
#include "rdsp_sync.h"

struct RdspSpecParams {
  const uint32_t *iq; /* [ch][in_stride] I | Q << 16 */
  size_t in_stride;
  int n_blocks;
  int have_prev; /* 0: the very first block only primes prevblock (FFTIQ.cpp:73-77) */
  int count0;    /* `count` at entry (FFTIQ.h:105) */
  int naverage;
  /* x / naverage without the ~30-instruction 32-bit division (4 per frame and lane): power of
   * two -> shift; else q = mulhi(div_magic, x) >> div_shift with div_magic = ceil(2^(31+L) / d),
   * L = ceil(log2 d), div_shift = L - 1.  Exact for every x < 2^31 (the error term x e / (d 2^(31+L))
   * with e < d <= 2^L stays below 1/d for x <= 2^31), and re^2 + im^2 <= 2 * 32768^2 = 2^31. */
  uint32_t div_magic;
  int div_shift, div_pow2;
  int use_window;
  const int16_t *window;  /* [256] q15 */
  const uint32_t *twid;   /* [192] cos | sin << 16 of 2 pi m / 256 (rdsp_q15_twiddles) */
  const uint16_t *sqrt_guess; /* [33] */
  uint32_t *st_prev;      /* [ch][128] previous block */
  uint32_t *st_sum;       /* [ch][256] sum[], position order */
  uint16_t *out;          /* [ch][out_stride][256] */
  size_t out_stride;      /* spectra per channel row */
};

namespace {
using namespace rdsp_q15;

__global__ void __launch_bounds__(64) rdsp_spectrum_kernel(RdspSpecParams p) {
  __shared__ uint32_t ex[256];
  const int t = threadIdx.x;
  const size_t ch = blockIdx.x;
  const uint32_t *iq = p.iq + ch * p.in_stride;

  /* per-lane constants: twiddles of stages 1..3 (output k = 0 and the whole of stage 4 take none),
   * window taps doubled (see window_mul) */
  Twiddle tw[3][4];
#pragma unroll
  for (int st = 0; st < 3; st++) {
    const int L = 64 >> (2 * st);
    const int j = t % L;
#pragma unroll
    for (int k = 1; k < 4; k++) tw[st][k] = make_twiddle(p.twid[k * j * (64 / L)]);
  }
  int win2[4] = {0, 0, 0, 0};
  if (p.use_window) {
#pragma unroll
    for (int k = 0; k < 4; k++) win2[k] = 2 * (int)p.window[t + 64 * k];
  }
  /* bin of the position 4t + k this lane ends with: base-4 digit reversal */
  const int rbase = ((t & 3) << 4) | (((t >> 2) & 3) << 2) | ((t >> 4) & 3);

  uint32_t prev0 = p.st_prev[ch * 128 + t], prev1 = p.st_prev[ch * 128 + 64 + t];
  int count = p.count0;
  uint32_t sum[4];
#pragma unroll
  for (int k = 0; k < 4; k++) sum[k] = count ? p.st_sum[ch * 256 + 4 * t + k] : 0u; /* count == 0 restarts the sums, FFTIQ.cpp:88-93 */
  int n_out = 0;
  int b = 0;
  if (!p.have_prev && p.n_blocks > 0) { /* FFTIQ.cpp:73-77 */
    prev0 = iq[t];
    prev1 = iq[64 + t];
    b = 1;
  }
  uint32_t cur0 = 0, cur1 = 0;
  if (b < p.n_blocks) { cur0 = iq[(size_t)b * 128 + t]; cur1 = iq[(size_t)b * 128 + 64 + t]; }
  for (; b < p.n_blocks; b++) {
    uint32_t x[4] = {prev0, prev1, cur0, cur1}; /* copy_to_fft_buffer, FFTIQ.cpp:78-79 */
    prev0 = cur0;                                /* FFTIQ.cpp:116-117 */
    prev1 = cur1;
    if (b + 1 < p.n_blocks) { /* next block's words land while this frame computes */
      cur0 = iq[(size_t)(b + 1) * 128 + t];
      cur1 = iq[(size_t)(b + 1) * 128 + 64 + t];
    }
    if (p.use_window) { /* FFTIQ.cpp:50-63 */
#pragma unroll
      for (int k = 0; k < 4; k++) x[k] = window_mul(x[k], win2[k]);
    }
    /* four stages, span L = 64, 16, 4, 1; positions base + k*L */
    bfly<kFirstStage>(x, tw[0]);
#pragma unroll
    for (int k = 0; k < 4; k++) ex[t + 64 * k] = x[k];
    wg_sync<1>();
    {
      const int base = (t / 16) * 64 + (t % 16);
#pragma unroll
      for (int k = 0; k < 4; k++) x[k] = ex[base + 16 * k];
      bfly<kMiddleStage>(x, tw[1]);
      wg_sync<1>();
#pragma unroll
      for (int k = 0; k < 4; k++) ex[base + 16 * k] = x[k];
    }
    wg_sync<1>();
    {
      const int base = (t / 4) * 16 + (t % 4);
#pragma unroll
      for (int k = 0; k < 4; k++) x[k] = ex[base + 4 * k];
      bfly<kMiddleStage>(x, tw[2]);
      wg_sync<1>();
#pragma unroll
      for (int k = 0; k < 4; k++) ex[base + 4 * k] = x[k];
    }
    wg_sync<1>();
#pragma unroll
    for (int k = 0; k < 4; k++) x[k] = ex[4 * t + k];
    bfly<kLastStage>(x, nullptr);
    wg_sync<1>();
    /* FFTIQ.cpp:86-98 */
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t msq = magsq(x[k]);
      uint32_t term; /* magsq / naverage, FFTIQ.cpp:90 */
      if (p.div_pow2) {
        term = msq >> p.div_shift;
      } else {
        term = __umulhi(p.div_magic, msq) >> p.div_shift;
      }
      sum[k] += term;
    }
    count = (count + 1) & 255; /* `uint8_t count`, FFTIQ.h:105 */
    if (count == 0) { /* wrapped (averageTogether set below the running count): the next frame finds count == 0
                         and restarts the sums without an output, FFTIQ.cpp:88-93 */
#pragma unroll
      for (int k = 0; k < 4; k++) sum[k] = 0u;
    }
    if (count == p.naverage) { /* FFTIQ.cpp:99-113 */
      count = 0;
      uint16_t *o = p.out + (ch * p.out_stride + (size_t)n_out) * 256;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int bin = 64 * k + rbase;
        o[255 - (bin ^ 128)] = (uint16_t)sqrt_uint32_approx(sum[k], p.sqrt_guess);
        sum[k] = 0u;
      }
      n_out++;
    }
  }
  p.st_prev[ch * 128 + t] = prev0;
  p.st_prev[ch * 128 + 64 + t] = prev1;
#pragma unroll
  for (int k = 0; k < 4; k++) p.st_sum[ch * 256 + 4 * t + k] = sum[k];
}
}  // namespace

/* ---- host side -------------------------------------------------------------- */
struct rdsp_spectrum {
  int n_channels, device;
  int naverage;
  int has_window; /* `const int16_t *window` non-NULL, FFTIQ.h:101, FFTIQ.cpp:81 */
  int have_prev, count;
  int16_t *d_window = nullptr;
  uint16_t *d_guess = nullptr;
  uint32_t *d_twid = nullptr, *d_prev = nullptr, *d_sum = nullptr;
};

#define SPEC_TRY(expr)                                                          \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      rdsp_set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
      return RDSP_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)

/* windowFunction(const int16_t *w), FFTIQ.h:93-95: the analyser keeps the caller's table (a copy
 * here: the table lives in device memory); NULL switches the window off (FFTIQ.cpp:81) */
static int upload_window(rdsp_spectrum_t *s, const int16_t *w256) {
  s->has_window = w256 != nullptr;
  if (w256) SPEC_TRY(hipMemcpy(s->d_window, w256, 256 * sizeof(int16_t), hipMemcpyHostToDevice));
  return RDSP_OK;
}
static int upload_window_id(rdsp_spectrum_t *s, int window_id) {
  int16_t w[256];
  if (window_id == RDSP_WINDOW_NONE) return upload_window(s, nullptr);
  if (window_id < 0 || window_id > RDSP_WINDOW_TUKEY) {
    rdsp_set_error("unknown window id %d", window_id);
    return RDSP_ERR_INVALID;
  }
  rdsp_window_q15(window_id, w);
  return upload_window(s, w);
}

/* AudioAnalyzeFFT256IQ() with explicit settings (the constructor's own are rdsp_spectrum_create_default) */
extern "C" int rdsp_spectrum_create(int n_channels, int device, int naverage, int window_id,
                                    rdsp_spectrum_t **out) {
  if (!out || n_channels <= 0 || naverage > 255) {
    rdsp_set_error("rdsp_spectrum_create: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (rdsp_device_count() <= 0) {
    rdsp_set_error("no HIP device: the rdsp product path has no CPU fallback");
    return RDSP_ERR_NO_DEVICE;
  }
  rdsp_spectrum_t *s = new rdsp_spectrum();
  s->n_channels = n_channels;
  s->device = device;
  s->naverage = naverage <= 0 ? 1 : naverage; /* averageTogether, FFTIQ.h:88-91 */
  s->has_window = 0;
  s->have_prev = 0;
  s->count = 0;
  uint32_t tw[192];
  rdsp_q15_twiddles(256, tw);
  auto setup = [&]() -> int { /* on any failure the half-made object is destroyed below, not leaked */
    SPEC_TRY(hipSetDevice(device));
    SPEC_TRY(hipMalloc((void **)&s->d_window, 256 * sizeof(int16_t)));
    SPEC_TRY(hipMalloc((void **)&s->d_twid, 192 * sizeof(uint32_t)));
    SPEC_TRY(hipMalloc((void **)&s->d_guess, 33 * sizeof(uint16_t)));
    SPEC_TRY(hipMalloc((void **)&s->d_prev, (size_t)n_channels * 128 * sizeof(uint32_t)));
    SPEC_TRY(hipMalloc((void **)&s->d_sum, (size_t)n_channels * 256 * sizeof(uint32_t)));
    SPEC_TRY(hipMemset(s->d_prev, 0, (size_t)n_channels * 128 * sizeof(uint32_t)));
    SPEC_TRY(hipMemset(s->d_sum, 0, (size_t)n_channels * 256 * sizeof(uint32_t)));
    SPEC_TRY(hipMemcpy(s->d_twid, tw, sizeof(tw), hipMemcpyHostToDevice));
    SPEC_TRY(hipMemcpy(s->d_guess, rdsp_sqrt_guess_table(), 33 * sizeof(uint16_t), hipMemcpyHostToDevice));
    return RDSP_OK;
  };
  if (setup() != RDSP_OK) {
    rdsp_spectrum_destroy(s);
    return RDSP_ERR_HIP;
  }
  int rc = upload_window_id(s, window_id);
  if (rc != RDSP_OK) {
    rdsp_spectrum_destroy(s);
    return rc;
  }
  *out = s;
  return RDSP_OK;
}
/* AudioAnalyzeFFT256IQ(), FFTIQ.h:55-60: window(AudioWindowBlackmanNuttall256), naverage(8) */
extern "C" int rdsp_spectrum_create_default(int n_channels, int device, rdsp_spectrum_t **out) {
  return rdsp_spectrum_create(n_channels, device, 8, RDSP_WINDOW_BLACKMAN_NUTTALL, out);
}

extern "C" int rdsp_spectrum_device(const rdsp_spectrum_t *s) { return s ? s->device : -1; }
extern "C" void rdsp_spectrum_destroy(rdsp_spectrum_t *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  (void)hipFree(s->d_window);
  (void)hipFree(s->d_twid);
  (void)hipFree(s->d_guess);
  (void)hipFree(s->d_prev);
  (void)hipFree(s->d_sum);
  delete s;
}

extern "C" int rdsp_spectrum_averageTogether(rdsp_spectrum_t *s, int n) { /* FFTIQ.h:88-91 */
  if (!s || n > 255) return RDSP_ERR_INVALID;
  s->naverage = n <= 0 ? 1 : n;
  return RDSP_OK;
}
extern "C" int rdsp_spectrum_windowFunction(rdsp_spectrum_t *s, int window_id) { /* FFTIQ.h:93-95 */
  if (!s) return RDSP_ERR_INVALID;
  SPEC_TRY(hipSetDevice(s->device));
  SPEC_TRY(hipDeviceSynchronize());
  return upload_window_id(s, window_id);
}
/* void windowFunction(const int16_t *w), FFTIQ.h:93-95, with the reference's own argument: a
 * host pointer to 256 q15 taps (e.g. AudioWindowHanning256, INO:144), or NULL for no window */
extern "C" int rdsp_spectrum_windowFunction_table(rdsp_spectrum_t *s, const int16_t *w256) {
  if (!s) return RDSP_ERR_INVALID;
  SPEC_TRY(hipSetDevice(s->device));
  SPEC_TRY(hipDeviceSynchronize());
  return upload_window(s, w256);
}

/* the reference's frame counter over `frames` frames: `if (++count == naverage) { output; count = 0; }`
 * with `uint8_t count` (FFTIQ.h:105, FFTIQ.cpp:99-100) -- an averageTogether() below the running count
 * lets it run on to the 8-bit wrap, exactly as on the Teensy */
static void advance_count(int *count, int naverage, int frames, int *outputs) {
  int c = *count, n = 0;
  if (c < naverage) { /* the usual case in closed form */
    const long long t = (long long)c + frames;
    n = (int)(t / naverage);
    c = (int)(t % naverage);
  } else {
    for (int f = 0; f < frames; f++) {
      c = (c + 1) & 255;
      if (c == naverage) { n++; c = 0; }
      if (c < naverage) { /* back in the usual regime: finish in closed form */
        const long long t = (long long)c + (frames - f - 1);
        n += (int)(t / naverage);
        c = (int)(t % naverage);
        break;
      }
    }
  }
  *count = c;
  if (outputs) *outputs = n;
}

/* number of spectra the next update over n_blocks will produce */
extern "C" int rdsp_spectrum_outputs_for(const rdsp_spectrum_t *s, int n_blocks) {
  if (!s || n_blocks <= 0) return 0;
  const int frames = n_blocks - (s->have_prev ? 0 : 1);
  int count = s->count, n = 0;
  advance_count(&count, s->naverage, frames, &n);
  return n;
}

/* n_blocks update() ticks for every channel (FFTIQ.cpp:65-118).  d_iq: int16
 * [n_channels][in_stride][2]; d_out: uint16 [n_channels][out_stride][256] receives
 * the spectra completed in this call (`available()` became true that many times). */
extern "C" int rdsp_spectrum_update(rdsp_spectrum_t *s, const int16_t *d_iq, size_t in_stride,
                                    int n_blocks, uint16_t *d_out, size_t out_stride,
                                    int *n_outputs, void *stream) {
  if (!s || !d_iq || n_blocks <= 0 || in_stride < (size_t)n_blocks * 128) {
    rdsp_set_error("rdsp_spectrum_update: bad argument");
    return RDSP_ERR_INVALID;
  }
  const int nout = rdsp_spectrum_outputs_for(s, n_blocks);
  if (nout > 0 && (!d_out || out_stride < (size_t)nout)) {
    rdsp_set_error("rdsp_spectrum_update: output buffer holds %zu spectra per channel, %d needed", out_stride, nout);
    return RDSP_ERR_INVALID;
  }
  SPEC_TRY(hipSetDevice(s->device));
  RdspSpecParams p;
  memset(&p, 0, sizeof(p));
  p.iq = reinterpret_cast<const uint32_t *>(d_iq);
  p.in_stride = in_stride;
  p.n_blocks = n_blocks;
  p.have_prev = s->have_prev;
  p.count0 = s->count;
  p.naverage = s->naverage;
  {
    const uint32_t d = (uint32_t)s->naverage;
    int fl = 0;
    while ((2u << fl) <= d) fl++; /* floor(log2 d) */
    p.div_pow2 = (d & (d - 1)) == 0;
    p.div_shift = fl; /* power of two: the shift itself; else L - 1 with L = fl + 1 */
    p.div_magic = 0;
    if (!p.div_pow2) {
      const unsigned long long n = 1ull << (31 + fl + 1);
      p.div_magic = (uint32_t)((n + d - 1) / d); /* < 2^32 because d > 2^fl */
    }
  }
  p.use_window = s->has_window;
  p.window = s->d_window;
  p.twid = s->d_twid;
  p.sqrt_guess = s->d_guess;
  p.st_prev = s->d_prev;
  p.st_sum = s->d_sum;
  p.out = d_out;
  p.out_stride = out_stride;
  hipLaunchKernelGGL(rdsp_spectrum_kernel, dim3(s->n_channels), dim3(64), 0, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rdsp_set_error("spectrum kernel launch failed: %s", hipGetErrorString(e));
    return RDSP_ERR_HIP;
  }
  const int frames = n_blocks - (s->have_prev ? 0 : 1);
  advance_count(&s->count, s->naverage, frames, nullptr);
  s->have_prev = 1;
  if (n_outputs) *n_outputs = nout;
  return RDSP_OK;
}
