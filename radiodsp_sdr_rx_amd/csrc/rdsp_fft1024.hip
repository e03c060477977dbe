/*
 * rdsp_fft1024.hip -- AudioAnalyzeFFT1024 batched over channels (the second analyser of the
 * sketch's graph: `AudioAnalyzeFFT1024 AudioFFT;` fed from Q_out_L, RadioDSP_SDR_RX.ino:57,87,
 * read by the display).  The Teensy Audio library is not in the tree; its update() is restated
 * from the library's published behaviour: blocks are collected eight at a time and four
 * are kept (1024-sample frames, hop 512), the samples are real (imaginary parts zero), q15 window
 * (x*w) >> 15, arm_cfft_radix4_q15 as CMSIS publishes it (rdsp_q15.h),
 * output[i] = sqrt_uint32_approx(re^2 + im^2) for bins 0..511; window (AudioWindowHanning1024,
 * INO:147), twiddles and the square root's guess table are those of the reference's firmware
 * image (rdsp_q15_tables.c).  Integer arithmetic: bit-exact against the CPU restatement kept
 * with the tests.
 *
 * One wave per channel.  Lane t runs the four butterflies b = t + 64 m of each of the five
 * stages (span L = 256, 64, 16, 4, 1); the packed int16 pairs go through a 4 KiB LDS buffer
 * between stages, the twiddle table W_1024^m sits in LDS too.  A frame reads 2 KiB of int16.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "rdsp_host.h"
#include "rdsp_q15.h"
#include "rdsp_sync.h"

struct RdspFft1024Params {
  const int16_t *audio; /* [ch][in_stride] samples, every in_step int16 */
  size_t in_stride;
  int in_step;
  int n_new;            /* new samples in this call (n_blocks * 128) */
  int have;             /* samples already buffered per channel (0 .. 7*128), oldest first */
  int16_t *st_hist;     /* [ch][896] buffered samples */
  int n_frames;         /* frames completed by this call */
  int use_window;
  const int16_t *window;  /* [1024] q15 */
  const uint32_t *twid;   /* [768] cos | sin << 16 of 2 pi m / 1024 (rdsp_q15_twiddles) */
  const uint16_t *sqrt_guess; /* [33] */
  uint16_t *out;          /* [ch][out_stride][512] */
  size_t out_stride;
  int keep;             /* samples to buffer after this call */
};

namespace {
using namespace rdsp_q15;

/* sample `idx` of the channel's stream = [buffered history | this call's new samples] */
__device__ __forceinline__ int sample_at(const RdspFft1024Params &p, size_t ch, int idx) {
  if (idx < p.have) return p.st_hist[ch * 896 + idx];
  return p.audio[(ch * p.in_stride + (size_t)(idx - p.have)) * p.in_step];
}

__global__ void __launch_bounds__(64) rdsp_fft1024_kernel(RdspFft1024Params p) {
  __shared__ uint32_t ex[1024];
  __shared__ Twiddle tws[768]; /* W_1024^m as the two dot-product operands (rdsp_q15.h) */
  const int t = threadIdx.x;
  const size_t ch = blockIdx.x;
  for (int i = t; i < 768; i += 64) tws[i] = make_twiddle(p.twid[i]);
  wg_sync<1>();

  for (int f = 0; f < p.n_frames; f++) {
    const int base = 512 * f;
    /* stage 1 (span 256): butterfly b = t + 64 m takes positions b + 256 k straight from the
     * stream: copy_to_fft_buffer + window + the first radix-4 pass */
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int b = t + 64 * m;
      uint32_t x[4];
      Twiddle tw[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int pos = b + 256 * k;
        int v = sample_at(p, ch, base + pos);
        if (p.use_window) v = __mul24(v, (int)p.window[pos]) >> 15;
        x[k] = pack16(v, 0);
        if (k) tw[k] = tws[k * b]; /* j = b, n / (4 L) = 1; output 0 takes no twiddle */
      }
      bfly<kFirstStage>(x, tw);
#pragma unroll
      for (int k = 0; k < 4; k++) ex[b + 256 * k] = x[k];
    }
    wg_sync<1>();
    /* stages 2..5: span L = 64, 16, 4, 1 */
#pragma unroll
    for (int st = 1; st < 5; st++) {
      const int L = 256 >> (2 * st);
      uint32_t y[4][4];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const int b = t + 64 * m;
        const int g = (b / L) * 4 * L, j = b % L;
        Twiddle tw[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          y[m][k] = ex[g + j + k * L];
          if (k && L > 1) tw[k] = tws[k * j * (256 / L)];
        }
        if (L > 1) bfly<kMiddleStage>(y[m], tw);
        else bfly<kLastStage>(y[m], tw); /* no twiddles */
      }
      wg_sync<1>();
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const int b = t + 64 * m;
        const int g = (b / L) * 4 * L, j = b % L;
#pragma unroll
        for (int k = 0; k < 4; k++) ex[g + j + k * L] = y[m][k];
      }
      wg_sync<1>();
    }
    /* position q holds bin digit-reverse_4(q) (five base-4 digits); bins 0..511 go out */
    uint16_t *o = p.out + (ch * p.out_stride + (size_t)f) * 512;
    for (int q = t; q < 1024; q += 64) {
      int bin = 0, r = q;
#pragma unroll
      for (int d = 0; d < 5; d++) { bin = (bin << 2) | (r & 3); r >>= 2; }
      if (bin < 512) {
        o[bin] = (uint16_t)sqrt_uint32_approx(magsq(ex[q]), p.sqrt_guess);
      }
    }
    wg_sync<1>();
  }
  /* the last `keep` samples of the stream stay buffered for the next call */
  const int total = p.have + p.n_new;
  int16_t tmp[14];
  int n_tmp = 0;
  for (int i = t; i < p.keep; i += 64) tmp[n_tmp++] = (int16_t)sample_at(p, ch, total - p.keep + i);
  wg_sync<1>(); /* every lane has read the old history before anyone overwrites it */
  n_tmp = 0;
  for (int i = t; i < p.keep; i += 64) p.st_hist[ch * 896 + i] = tmp[n_tmp++];
}
}  // namespace

struct rdsp_fft1024 {
  int n_channels, device;
  int has_window = 0;
  int have = 0; /* samples buffered per channel */
  int16_t *d_window = nullptr, *d_hist = nullptr;
  uint16_t *d_guess = nullptr;
  uint32_t *d_twid = nullptr;
};

#define F1K_TRY(expr)                                                           \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      rdsp_set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
      return RDSP_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)

static int fft1024_upload_window(rdsp_fft1024_t *s, const int16_t *w1024) {
  s->has_window = w1024 != nullptr;
  if (w1024) F1K_TRY(hipMemcpy(s->d_window, w1024, 1024 * sizeof(int16_t), hipMemcpyHostToDevice));
  return RDSP_OK;
}
static int fft1024_upload_window_id(rdsp_fft1024_t *s, int window_id) {
  if (window_id == RDSP_WINDOW_NONE) return fft1024_upload_window(s, nullptr);
  if (window_id < 0 || window_id > RDSP_WINDOW_TUKEY) {
    rdsp_set_error("unknown window id %d", window_id);
    return RDSP_ERR_INVALID;
  }
  std::vector<int16_t> w(1024);
  rdsp_window_q15_n(window_id, 1024, w.data());
  return fft1024_upload_window(s, w.data());
}

extern "C" int rdsp_fft1024_create(int n_channels, int device, int window_id, rdsp_fft1024_t **out) {
  if (!out || n_channels <= 0) {
    rdsp_set_error("rdsp_fft1024_create: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (rdsp_device_count() <= 0) {
    rdsp_set_error("no HIP device: the rdsp product path has no CPU fallback");
    return RDSP_ERR_NO_DEVICE;
  }
  rdsp_fft1024_t *s = new rdsp_fft1024();
  s->n_channels = n_channels;
  s->device = device;
  std::vector<uint32_t> tw(768);
  rdsp_q15_twiddles(1024, tw.data());
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&s->d_window, 1024 * sizeof(int16_t)) != hipSuccess ||
      hipMalloc((void **)&s->d_twid, 768 * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void **)&s->d_guess, 33 * sizeof(uint16_t)) != hipSuccess ||
      hipMalloc((void **)&s->d_hist, (size_t)n_channels * 896 * sizeof(int16_t)) != hipSuccess ||
      hipMemset(s->d_hist, 0, (size_t)n_channels * 896 * sizeof(int16_t)) != hipSuccess ||
      hipMemcpy(s->d_twid, tw.data(), 768 * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->d_guess, rdsp_sqrt_guess_table(), 33 * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess ||
      fft1024_upload_window_id(s, window_id) != RDSP_OK) {
    rdsp_set_error("rdsp_fft1024_create: device set-up failed");
    rdsp_fft1024_destroy(s);
    return RDSP_ERR_HIP;
  }
  *out = s;
  return RDSP_OK;
}
extern "C" void rdsp_fft1024_destroy(rdsp_fft1024_t *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  if (s->d_window) (void)hipFree(s->d_window);
  if (s->d_twid) (void)hipFree(s->d_twid);
  if (s->d_guess) (void)hipFree(s->d_guess);
  if (s->d_hist) (void)hipFree(s->d_hist);
  delete s;
}
extern "C" int rdsp_fft1024_windowFunction(rdsp_fft1024_t *s, int window_id) {
  if (!s) return RDSP_ERR_INVALID;
  F1K_TRY(hipSetDevice(s->device));
  F1K_TRY(hipDeviceSynchronize());
  return fft1024_upload_window_id(s, window_id);
}
/* windowFunction(const int16_t *w) with the library's own argument: a host pointer to 1024 q15 taps
 * (AudioWindowHanning1024, INO:147) or NULL */
extern "C" int rdsp_fft1024_windowFunction_table(rdsp_fft1024_t *s, const int16_t *w1024) {
  if (!s) return RDSP_ERR_INVALID;
  F1K_TRY(hipSetDevice(s->device));
  F1K_TRY(hipDeviceSynchronize());
  return fft1024_upload_window(s, w1024);
}
/* AudioFFT.averageTogether(30) (INO:148): the library's 1024-point analyser declares it and does
 * nothing with it ("not implemented yet"); accepted and ignored here too */
extern "C" int rdsp_fft1024_averageTogether(rdsp_fft1024_t *s, int n) {
  (void)n;
  return s ? RDSP_OK : RDSP_ERR_INVALID;
}
/* frames the next update over n_blocks completes: the first after 8 blocks, then one per 4 */
extern "C" int rdsp_fft1024_outputs_for(const rdsp_fft1024_t *s, int n_blocks) {
  if (!s || n_blocks <= 0) return 0;
  const int total = s->have / 128 + n_blocks;
  return total < 8 ? 0 : (total - 8) / 4 + 1;
}
extern "C" int rdsp_fft1024_update(rdsp_fft1024_t *s, const int16_t *d_audio, size_t in_stride, int in_step, int n_blocks,
                                   uint16_t *d_out, size_t out_stride, int *n_outputs, void *stream) {
  if (!s || !d_audio || n_blocks <= 0 || in_step < 1 || in_stride < (size_t)n_blocks * 128) {
    rdsp_set_error("rdsp_fft1024_update: bad argument");
    return RDSP_ERR_INVALID;
  }
  const int nf = rdsp_fft1024_outputs_for(s, n_blocks);
  if (nf > 0 && (!d_out || out_stride < (size_t)nf)) {
    rdsp_set_error("rdsp_fft1024_update: output buffer holds %zu spectra per channel, %d needed", out_stride, nf);
    return RDSP_ERR_INVALID;
  }
  F1K_TRY(hipSetDevice(s->device));
  const int total = s->have + n_blocks * 128;
  RdspFft1024Params p;
  memset(&p, 0, sizeof(p));
  p.audio = d_audio;
  p.in_stride = in_stride;
  p.in_step = in_step;
  p.n_new = n_blocks * 128;
  p.have = s->have;
  p.st_hist = s->d_hist;
  p.n_frames = nf;
  p.use_window = s->has_window;
  p.window = s->d_window;
  p.twid = s->d_twid;
  p.sqrt_guess = s->d_guess;
  p.out = d_out;
  p.out_stride = out_stride;
  p.keep = nf > 0 ? total - 512 * nf : total; /* 4..7 blocks once frames run, everything before */
  hipLaunchKernelGGL(rdsp_fft1024_kernel, dim3(s->n_channels), dim3(64), 0, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    rdsp_set_error("fft1024 kernel launch failed: %s", hipGetErrorString(e));
    return RDSP_ERR_HIP;
  }
  s->have = p.keep;
  if (n_outputs) *n_outputs = nf;
  return RDSP_OK;
}

/* ---- the node: one input (Q_out_L in the sketch), no outputs ------------------------------ */
namespace {
struct Fft1024Node {
  rdsp_fft1024_t *an;
  int n_channels;
  std::vector<uint16_t> h_out; /* [ch][512] */
  int16_t *d_in = nullptr;
  uint16_t *d_out = nullptr;
  hipStream_t stream = nullptr;
  int outputflag = 0, status = RDSP_OK;
  int device = 0; /* the object's device: selected in update and destroy (a process may drive several GPUs) */
};
void fft1024_node_destroy(void *u) {
  Fft1024Node *s = static_cast<Fft1024Node *>(u);
  (void)hipSetDevice(s->device); /* the node's buffers and stream live on its object's device */
  if (s->d_in) (void)hipFree(s->d_in);
  if (s->d_out) (void)hipFree(s->d_out);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}
void fft1024_node_update(rdsp_node_t *n, void *u) {
  Fft1024Node *s = static_cast<Fft1024Node *>(u);
  rdsp_block_t *b = rdsp_receive_readonly(n, 0);
  if (!b) return;
  (void)hipSetDevice(s->device);
  const size_t bytes = (size_t)s->n_channels * RDSP_BLOCK_SAMPLES * sizeof(int16_t);
  int n_out = 0;
  hipError_t e = hipMemcpyAsync(s->d_in, rdsp_block_data(b), bytes, hipMemcpyHostToDevice, s->stream);
  int rc = RDSP_OK;
  if (e == hipSuccess) rc = rdsp_fft1024_update(s->an, s->d_in, RDSP_BLOCK_SAMPLES, 1, 1, s->d_out, 1, &n_out, s->stream);
  if (e == hipSuccess && rc == RDSP_OK && n_out > 0)
    e = hipMemcpyAsync(s->h_out.data(), s->d_out, s->h_out.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess && rc == RDSP_OK) e = hipStreamSynchronize(s->stream); /* the block is released below */
  rdsp_release(b);
  if (e != hipSuccess || rc != RDSP_OK) {
    s->status = (rc != RDSP_OK) ? rc : RDSP_ERR_HIP;
    if (e != hipSuccess) rdsp_set_error("fft1024 node: %s", hipGetErrorString(e));
    return;
  }
  if (n_out > 0) s->outputflag = 1;
}
}  // namespace

extern "C" rdsp_node_t *rdsp_fft1024_node_create(rdsp_graph_t *g, rdsp_fft1024_t *an) {
  if (!g || !an || an->n_channels != rdsp_graph_channels(g)) {
    rdsp_set_error("rdsp_fft1024_node_create: bad argument (the analyser needs the graph's channel count)");
    return nullptr;
  }
  Fft1024Node *s = new Fft1024Node();
  s->an = an;
  s->n_channels = an->n_channels;
  s->device = an->device;
  s->h_out.assign((size_t)s->n_channels * 512, 0);
  if (hipSetDevice(an->device) != hipSuccess ||
      hipMalloc((void **)&s->d_in, (size_t)s->n_channels * RDSP_BLOCK_SAMPLES * sizeof(int16_t)) != hipSuccess ||
      hipMalloc((void **)&s->d_out, s->h_out.size() * sizeof(uint16_t)) != hipSuccess ||
      hipStreamCreate(&s->stream) != hipSuccess) {
    rdsp_set_error("rdsp_fft1024_node_create: device allocation failed");
    fft1024_node_destroy(s);
    return nullptr;
  }
  rdsp_node_t *n = rdsp_node_create(g, 1, fft1024_node_update, s);
  if (!n) { fft1024_node_destroy(s); return nullptr; }
  rdsp_node_set_destructor(n, fft1024_node_destroy);
  return n;
}
extern "C" int rdsp_fft1024_node_available(rdsp_node_t *n) {
  Fft1024Node *s = static_cast<Fft1024Node *>(rdsp_node_user(n));
  if (!s) return 0;
  const int f = s->outputflag;
  s->outputflag = 0;
  return f;
}
extern "C" const uint16_t *rdsp_fft1024_node_output(rdsp_node_t *n) {
  Fft1024Node *s = static_cast<Fft1024Node *>(rdsp_node_user(n));
  return s ? s->h_out.data() : nullptr;
}
extern "C" float rdsp_fft1024_node_read(rdsp_node_t *n, int ch, unsigned int binNumber) {
  Fft1024Node *s = static_cast<Fft1024Node *>(rdsp_node_user(n));
  if (!s || ch < 0 || ch >= s->n_channels) return 0.0f;
  return rdsp_fft1024_read(s->h_out.data() + (size_t)ch * 512, binNumber);
}
extern "C" float rdsp_fft1024_node_read_range(rdsp_node_t *n, int ch, unsigned int binFirst, unsigned int binLast) {
  Fft1024Node *s = static_cast<Fft1024Node *>(rdsp_node_user(n));
  if (!s || ch < 0 || ch >= s->n_channels) return 0.0f;
  return rdsp_fft1024_read_range(s->h_out.data() + (size_t)ch * 512, binFirst, binLast);
}
extern "C" int rdsp_fft1024_node_status(rdsp_node_t *n) {
  Fft1024Node *s = static_cast<Fft1024Node *>(rdsp_node_user(n));
  return s ? s->status : RDSP_ERR_INVALID;
}
