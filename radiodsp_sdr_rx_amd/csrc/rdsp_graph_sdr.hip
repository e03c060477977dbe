/*
 * rdsp_graph_sdr.hip -- the SDR engine node of the block graph: the role of
 * `AudioSDR SDR;` (+ the convolutional stage that loop() runs between the record
 * and play queues) in RadioDSP_SDR_RX.ino:53-54,81-89.  Two inputs (I, Q tiles),
 * two outputs (L, R tiles).  update() gathers input tiles until the chain's
 * granule is available (the `available() > N_BLOCKS` gate of
 * RDSP_convolutional.h:231), runs the GPU chain on them and hands the audio out
 * one 128-sample tile pair per tick.  Every compute step is rdsp_chain_process;
 * there is no host DSP here.
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <vector>

#include "rdsp_host.h"

namespace {
struct SdrNode {
  rdsp_chain_t *chain;
  int n_channels, gran, decim;
  int have; /* input blocks staged */
  std::vector<int16_t> h_iq;  /* [ch][gran*128][2] */
  std::vector<int16_t> h_out; /* [ch][gran*128/decim][2] */
  int16_t *d_iq = nullptr, *d_out = nullptr;
  hipStream_t stream = nullptr;
  int device = 0;
  std::deque<std::vector<int16_t>> out_l, out_r; /* audio tiles waiting for a tick */
  int status = RDSP_OK;
};

void sdr_destroy(void *u) {
  SdrNode *s = static_cast<SdrNode *>(u);
  (void)hipSetDevice(s->device);
  if (s->d_iq) (void)hipFree(s->d_iq);
  if (s->d_out) (void)hipFree(s->d_out);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

void sdr_update(rdsp_node_t *n, void *u) {
  SdrNode *s = static_cast<SdrNode *>(u);
  rdsp_block_t *bi = rdsp_receive_readonly(n, 0);
  rdsp_block_t *bq = rdsp_receive_readonly(n, 1);
  if (bi && bq) {
    const int16_t *pi = rdsp_block_data(bi), *pq = rdsp_block_data(bq);
    const size_t row = (size_t)s->gran * RDSP_BLOCK_SAMPLES;
    for (int c = 0; c < s->n_channels; c++) {
      int16_t *dst = &s->h_iq[((size_t)c * row + (size_t)s->have * RDSP_BLOCK_SAMPLES) * 2];
      const int16_t *si = pi + (size_t)c * RDSP_BLOCK_SAMPLES, *sq = pq + (size_t)c * RDSP_BLOCK_SAMPLES;
      for (int i = 0; i < RDSP_BLOCK_SAMPLES; i++) {
        dst[2 * i] = si[i];
        dst[2 * i + 1] = sq[i];
      }
    }
    s->have++;
  }
  rdsp_release(bi); /* a lone I or Q block is dropped, like a node returning early */
  rdsp_release(bq);

  if (s->have == s->gran) {
    const size_t in_row = (size_t)s->gran * RDSP_BLOCK_SAMPLES;
    const size_t out_row = in_row / (size_t)s->decim;
    hipError_t e = hipSetDevice(s->device);
    if (e == hipSuccess)
      e = hipMemcpyAsync(s->d_iq, s->h_iq.data(), s->h_iq.size() * sizeof(int16_t), hipMemcpyHostToDevice, s->stream);
    int rc = RDSP_OK;
    if (e == hipSuccess)
      rc = rdsp_chain_process(s->chain, s->d_iq, in_row, s->gran, s->d_out, out_row, nullptr, s->stream);
    /* a chain in pipelined mode finishes d_out on its internal tail stream: the copy below waits for it */
    if (e == hipSuccess && rc == RDSP_OK) rc = rdsp_chain_flush(s->chain, s->stream);
    if (e == hipSuccess && rc == RDSP_OK)
      e = hipMemcpyAsync(s->h_out.data(), s->d_out, s->h_out.size() * sizeof(int16_t),
                         hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess && rc == RDSP_OK) e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess || rc != RDSP_OK) {
      s->status = (rc != RDSP_OK) ? rc : RDSP_ERR_HIP;
      if (e != hipSuccess) rdsp_set_error("sdr node: %s", hipGetErrorString(e));
    } else {
      const int n_tiles = (int)(out_row / RDSP_BLOCK_SAMPLES);
      for (int t = 0; t < n_tiles; t++) {
        std::vector<int16_t> L((size_t)s->n_channels * RDSP_BLOCK_SAMPLES), R(L.size());
        for (int c = 0; c < s->n_channels; c++) {
          const int16_t *src = &s->h_out[((size_t)c * out_row + (size_t)t * RDSP_BLOCK_SAMPLES) * 2];
          for (int i = 0; i < RDSP_BLOCK_SAMPLES; i++) {
            L[(size_t)c * RDSP_BLOCK_SAMPLES + i] = src[2 * i];
            R[(size_t)c * RDSP_BLOCK_SAMPLES + i] = src[2 * i + 1];
          }
        }
        s->out_l.push_back(std::move(L));
        s->out_r.push_back(std::move(R));
      }
    }
    s->have = 0;
  }

  if (!s->out_l.empty()) {
    rdsp_block_t *bl = rdsp_allocate(n), *br = rdsp_allocate(n);
    if (bl && br) {
      memcpy(rdsp_block_data(bl), s->out_l.front().data(), s->out_l.front().size() * sizeof(int16_t));
      memcpy(rdsp_block_data(br), s->out_r.front().data(), s->out_r.front().size() * sizeof(int16_t));
      s->out_l.pop_front();
      s->out_r.pop_front();
      rdsp_transmit(n, bl, 0);
      rdsp_transmit(n, br, 1);
    }
    rdsp_release(bl);
    rdsp_release(br);
  }
}
}  // namespace

extern "C" rdsp_node_t *rdsp_sdr_node_create(rdsp_graph_t *g, rdsp_chain_t *chain) {
  if (!g || !chain || rdsp_graph_channels(g) != rdsp_chain_channels(chain)) {
    rdsp_set_error("rdsp_sdr_node_create: graph and chain must have the same channel count");
    return nullptr;
  }
  SdrNode *s = new SdrNode();
  s->chain = chain;
  s->n_channels = rdsp_chain_channels(chain);
  s->gran = rdsp_chain_call_unit_blocks(chain); /* one call per call unit: a fixed split, the same grid on every run */
  s->decim = rdsp_chain_decim(chain);
  s->have = 0;
  const size_t in_n = (size_t)s->n_channels * s->gran * RDSP_BLOCK_SAMPLES * 2;
  s->h_iq.assign(in_n, 0);
  s->h_out.assign(in_n / s->decim, 0);
  s->device = rdsp_chain_device(chain); /* the node's buffers and stream live where the chain does */
  if (hipSetDevice(s->device) != hipSuccess ||
      hipMalloc((void **)&s->d_iq, in_n * sizeof(int16_t)) != hipSuccess ||
      hipMalloc((void **)&s->d_out, in_n / s->decim * sizeof(int16_t)) != hipSuccess ||
      hipStreamCreate(&s->stream) != hipSuccess) {
    rdsp_set_error("rdsp_sdr_node_create: device allocation failed");
    sdr_destroy(s);
    return nullptr;
  }
  rdsp_node_t *n = rdsp_node_create(g, 2, sdr_update, s);
  if (!n) {
    sdr_destroy(s);
    return nullptr;
  }
  rdsp_node_set_destructor(n, sdr_destroy);
  return n;
}

extern "C" int rdsp_sdr_node_status(rdsp_node_t *n) {
  SdrNode *s = static_cast<SdrNode *>(rdsp_node_user(n));
  return s ? s->status : RDSP_ERR_INVALID;
}

/* ---- AudioAnalyzeFFT256IQ as a graph node (analyze_fft256iq.h:52-110) -----------------
 * `AudioAnalyzeFFT256IQ FFT;` with `AudioConnection(preProcessor, 0, FFT, 0)` and
 * (.., 1, FFT, 1) in the sketch (RadioDSP_SDR_RX.ino:57,73-74).  Two inputs (I, Q
 * tiles), no outputs; update() is FFTIQ.cpp:65-118 for every channel of the tile: the
 * tick's blocks are interleaved, uploaded and handed to rdsp_spectrum_update (which keeps
 * the previous block on the device); available()/output[] follow FFTIQ.h:62-73,99. */
namespace {
struct SpectrumNode {
  rdsp_spectrum_t *spec;
  int n_channels;
  std::vector<int16_t> h_iq;      /* [ch][128][2] */
  std::vector<uint16_t> h_out;    /* [ch][256]    */
  int16_t *d_iq = nullptr;
  uint16_t *d_out = nullptr;
  hipStream_t stream = nullptr;
  int device = 0;
  int outputflag = 0;             /* FFTIQ.h:63 */
  int status = RDSP_OK;
};

void spectrum_destroy(void *u) {
  SpectrumNode *s = static_cast<SpectrumNode *>(u);
  (void)hipSetDevice(s->device);
  if (s->d_iq) (void)hipFree(s->d_iq);
  if (s->d_out) (void)hipFree(s->d_out);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

void spectrum_update(rdsp_node_t *n, void *u) {
  SpectrumNode *s = static_cast<SpectrumNode *>(u);
  rdsp_block_t *bi = rdsp_receive_readonly(n, 0); /* FFTIQ.cpp:70-71 */
  rdsp_block_t *bq = rdsp_receive_readonly(n, 1);
  if (!bi || !bq) { /* FFTIQ.cpp:72: return when a block is missing */
    rdsp_release(bi);
    rdsp_release(bq);
    return;
  }
  const int16_t *pi = rdsp_block_data(bi), *pq = rdsp_block_data(bq);
  for (int c = 0; c < s->n_channels; c++) {
    int16_t *dst = &s->h_iq[(size_t)c * RDSP_BLOCK_SAMPLES * 2];
    for (int i = 0; i < RDSP_BLOCK_SAMPLES; i++) {
      dst[2 * i] = pi[(size_t)c * RDSP_BLOCK_SAMPLES + i];
      dst[2 * i + 1] = pq[(size_t)c * RDSP_BLOCK_SAMPLES + i];
    }
  }
  rdsp_release(bi); /* FFTIQ.cpp:114-115 (the previous block lives on the device) */
  rdsp_release(bq);
  int n_out = 0;
  hipError_t e = hipSetDevice(s->device);
  if (e == hipSuccess)
    e = hipMemcpyAsync(s->d_iq, s->h_iq.data(), s->h_iq.size() * sizeof(int16_t), hipMemcpyHostToDevice, s->stream);
  int rc = RDSP_OK;
  if (e == hipSuccess)
    rc = rdsp_spectrum_update(s->spec, s->d_iq, RDSP_BLOCK_SAMPLES, 1, s->d_out, 1, &n_out, s->stream);
  if (e == hipSuccess && rc == RDSP_OK && n_out > 0)
    e = hipMemcpyAsync(s->h_out.data(), s->d_out, s->h_out.size() * sizeof(uint16_t), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess && rc == RDSP_OK) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || rc != RDSP_OK) {
    s->status = (rc != RDSP_OK) ? rc : RDSP_ERR_HIP;
    if (e != hipSuccess) rdsp_set_error("spectrum node: %s", hipGetErrorString(e));
    return;
  }
  if (n_out > 0) s->outputflag = 1; /* FFTIQ.cpp:112 */
}
}  // namespace

extern "C" rdsp_node_t *rdsp_spectrum_node_create(rdsp_graph_t *g, rdsp_spectrum_t *spec) {
  if (!g || !spec) {
    rdsp_set_error("rdsp_spectrum_node_create: bad argument");
    return nullptr;
  }
  SpectrumNode *s = new SpectrumNode();
  s->spec = spec;
  s->n_channels = rdsp_graph_channels(g);
  s->h_iq.assign((size_t)s->n_channels * RDSP_BLOCK_SAMPLES * 2, 0);
  s->h_out.assign((size_t)s->n_channels * 256, 0);
  s->device = rdsp_spectrum_device(spec);
  if (hipSetDevice(s->device) != hipSuccess ||
      hipMalloc((void **)&s->d_iq, s->h_iq.size() * sizeof(int16_t)) != hipSuccess ||
      hipMalloc((void **)&s->d_out, s->h_out.size() * sizeof(uint16_t)) != hipSuccess ||
      hipStreamCreate(&s->stream) != hipSuccess) {
    rdsp_set_error("rdsp_spectrum_node_create: device allocation failed");
    spectrum_destroy(s);
    return nullptr;
  }
  rdsp_node_t *n = rdsp_node_create(g, 2, spectrum_update, s);
  if (!n) {
    spectrum_destroy(s);
    return nullptr;
  }
  rdsp_node_set_destructor(n, spectrum_destroy);
  return n;
}

/* FFTIQ.h:62-68: true once per finished average, cleared by the call */
extern "C" int rdsp_spectrum_node_available(rdsp_node_t *n) {
  SpectrumNode *s = static_cast<SpectrumNode *>(rdsp_node_user(n));
  if (!s) return 0;
  const int f = s->outputflag;
  s->outputflag = 0;
  return f;
}
/* FFTIQ.h:99 `uint16_t output[256]` of every channel: [n_channels][256], valid until the next update */
extern "C" const uint16_t *rdsp_spectrum_node_output(rdsp_node_t *n) {
  SpectrumNode *s = static_cast<SpectrumNode *>(rdsp_node_user(n));
  return s ? s->h_out.data() : nullptr;
}
/* FFT.read(bin) / FFT.read(binFirst, binLast) of channel `ch` (FFTIQ.h:70-86) */
extern "C" float rdsp_spectrum_node_read(rdsp_node_t *n, int ch, unsigned int binNumber) {
  SpectrumNode *s = static_cast<SpectrumNode *>(rdsp_node_user(n));
  if (!s || ch < 0 || ch >= s->n_channels) return 0.0f;
  return rdsp_spectrum_read(s->h_out.data() + (size_t)ch * 256, binNumber);
}
extern "C" float rdsp_spectrum_node_read_range(rdsp_node_t *n, int ch, unsigned int binFirst, unsigned int binLast) {
  SpectrumNode *s = static_cast<SpectrumNode *>(rdsp_node_user(n));
  if (!s || ch < 0 || ch >= s->n_channels) return 0.0f;
  return rdsp_spectrum_read_range(s->h_out.data() + (size_t)ch * 256, binFirst, binLast);
}
extern "C" int rdsp_spectrum_node_status(rdsp_node_t *n) {
  SpectrumNode *s = static_cast<SpectrumNode *>(rdsp_node_user(n));
  return s ? s->status : RDSP_ERR_INVALID;
}

/* ---- the reference's own engine objects as graph nodes -------------------------------------------------------------------
 * `AudioSDRpreProcessor preProcessor;` and `AudioSDR SDR;` (RadioDSP_SDR_RX.ino:53-54) wired as INO:71-72,81-86: two inputs
 * (I, Q tiles), two outputs, one block per tick like the library's update().  The arithmetic is rdsp_preproc_update /
 * rdsp_engine_update (csrc/rdsp_engine.hip); the node only carries tiles to the device and back. */
namespace {
struct PairNode {
  rdsp_engine_t *engine = nullptr;
  rdsp_preproc_t *pre = nullptr;
  int n_channels = 0, device = 0;
  std::vector<int16_t> h_in, h_out; /* [ch][128][2] */
  int16_t *d_in = nullptr, *d_out = nullptr;
  hipStream_t stream = nullptr;
  int status = RDSP_OK;
};
void pair_destroy(void *u) {
  PairNode *s = static_cast<PairNode *>(u);
  (void)hipSetDevice(s->device);
  if (s->d_in) (void)hipFree(s->d_in);
  if (s->d_out) (void)hipFree(s->d_out);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}
void pair_update(rdsp_node_t *n, void *u) {
  PairNode *s = static_cast<PairNode *>(u);
  rdsp_block_t *bi = rdsp_receive_readonly(n, 0), *bq = rdsp_receive_readonly(n, 1);
  if (!bi || !bq) { /* the library's update() returns when a block is missing (image 0xe756 ... 0xe77a, 0xeea4 ... 0xeebc) */
    rdsp_release(bi);
    rdsp_release(bq);
    return;
  }
  const int16_t *pi = rdsp_block_data(bi), *pq = rdsp_block_data(bq);
  for (int c = 0; c < s->n_channels; c++)
    for (int i = 0; i < RDSP_BLOCK_SAMPLES; i++) {
      s->h_in[((size_t)c * RDSP_BLOCK_SAMPLES + i) * 2] = pi[(size_t)c * RDSP_BLOCK_SAMPLES + i];
      s->h_in[((size_t)c * RDSP_BLOCK_SAMPLES + i) * 2 + 1] = pq[(size_t)c * RDSP_BLOCK_SAMPLES + i];
    }
  rdsp_release(bi);
  rdsp_release(bq);
  hipError_t e = hipSetDevice(s->device);
  if (e == hipSuccess) e = hipMemcpyAsync(s->d_in, s->h_in.data(), s->h_in.size() * sizeof(int16_t), hipMemcpyHostToDevice, s->stream);
  int rc = RDSP_OK;
  if (e == hipSuccess)
    rc = s->engine ? rdsp_engine_update(s->engine, s->d_in, RDSP_BLOCK_SAMPLES, 1, s->d_out, RDSP_BLOCK_SAMPLES, s->stream)
                   : rdsp_preproc_update(s->pre, s->d_in, RDSP_BLOCK_SAMPLES, 1, s->d_out, RDSP_BLOCK_SAMPLES, s->stream);
  if (e == hipSuccess && rc == RDSP_OK)
    e = hipMemcpyAsync(s->h_out.data(), s->d_out, s->h_out.size() * sizeof(int16_t), hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess && rc == RDSP_OK) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || rc != RDSP_OK) {
    s->status = rc != RDSP_OK ? rc : RDSP_ERR_HIP;
    if (e != hipSuccess) rdsp_set_error("engine node: %s", hipGetErrorString(e));
    return;
  }
  rdsp_block_t *b0 = rdsp_allocate(n), *b1 = rdsp_allocate(n);
  if (b0 && b1) {
    int16_t *o0 = rdsp_block_data(b0), *o1 = rdsp_block_data(b1);
    for (int c = 0; c < s->n_channels; c++)
      for (int i = 0; i < RDSP_BLOCK_SAMPLES; i++) {
        o0[(size_t)c * RDSP_BLOCK_SAMPLES + i] = s->h_out[((size_t)c * RDSP_BLOCK_SAMPLES + i) * 2];
        o1[(size_t)c * RDSP_BLOCK_SAMPLES + i] = s->h_out[((size_t)c * RDSP_BLOCK_SAMPLES + i) * 2 + 1];
      }
    rdsp_transmit(n, b0, 0);
    rdsp_transmit(n, b1, 1);
  }
  rdsp_release(b0);
  rdsp_release(b1);
}
rdsp_node_t *pair_create(rdsp_graph_t *g, rdsp_engine_t *engine, rdsp_preproc_t *pre) {
  const int nch = engine ? rdsp_engine_channels(engine) : rdsp_preproc_channels(pre);
  if (!g || rdsp_graph_channels(g) != nch) {
    rdsp_set_error("engine / pre-processor node: graph and object must have the same channel count");
    return nullptr;
  }
  PairNode *s = new PairNode();
  s->engine = engine; s->pre = pre; s->n_channels = nch;
  s->device = engine ? rdsp_engine_device(engine) : rdsp_preproc_device(pre);
  s->h_in.assign((size_t)nch * RDSP_BLOCK_SAMPLES * 2, 0);
  s->h_out.assign(s->h_in.size(), 0);
  if (hipSetDevice(s->device) != hipSuccess || hipMalloc((void **)&s->d_in, s->h_in.size() * sizeof(int16_t)) != hipSuccess ||
      hipMalloc((void **)&s->d_out, s->h_in.size() * sizeof(int16_t)) != hipSuccess || hipStreamCreate(&s->stream) != hipSuccess) {
    rdsp_set_error("engine / pre-processor node: device allocation failed");
    pair_destroy(s);
    return nullptr;
  }
  rdsp_node_t *n = rdsp_node_create(g, 2, pair_update, s);
  if (!n) { pair_destroy(s); return nullptr; }
  rdsp_node_set_destructor(n, pair_destroy);
  return n;
}
}  // namespace

extern "C" rdsp_node_t *rdsp_engine_node_create(rdsp_graph_t *g, rdsp_engine_t *e) { return e ? pair_create(g, e, nullptr) : nullptr; }
extern "C" rdsp_node_t *rdsp_preproc_node_create(rdsp_graph_t *g, rdsp_preproc_t *p) { return p ? pair_create(g, nullptr, p) : nullptr; }
extern "C" int rdsp_engine_node_status(rdsp_node_t *n) {
  PairNode *s = static_cast<PairNode *>(rdsp_node_user(n));
  return s ? s->status : RDSP_ERR_INVALID;
}
