/*
 * rdsp_biquad.hip -- biquad cascades on the GPU.
 *
 * Two callers share one kernel:
 *   * the engine's IIR audio filter bank: SDR.setAudioFilter(audioCW ... audioAM)
 *     (RDSP_controls.h:153-177) is a table of 8th-order band-passes in
 *     arm_biquad_cascade_df1_f32 layout in the shipped firmware; the chain
 *     runs it on the demodulated mono audio between the front kernel and the tail stage when
 *     rdsp_sdr_setAudioFilterKind selects it (rdsp_chain.hip);
 *   * AudioFilterBiquad nodes, `biquad1.setHighpass(0, 500, 0.5)` in front of the panadapter
 *     (RadioDSP_SDR_RX.ino:58-59,75-78,155-156): rdsp_biquad_t + rdsp_biquad_node_create.
 * Neither library is in the tree.  The arithmetic is arm_biquad_cascade_df1_f32's as CMSIS-DSP publishes it:
 * direct form 1 in float, per stage
 *     acc = (b0 * Xn) + (b1 * Xn1) + (b2 * Xn2) + (a1 * Yn1) + (a2 * Yn2)
 * summed left to right, EVERY PRODUCT ROUNDED BEFORE IT IS ADDED (feedback terms added, coefficient order
 * {b0, b1, b2, a1, a2}) -- the routine is in the reference's firmware image (the engine's audio filters run through
 * it) as 5 VMUL + 4 VADD per sample, no fused operation (tests/test_firmware_tables.py); until round 5 this was
 * a chain of four fused multiply-adds.  Four stages, unused ones pass through exactly.  Teensy's AudioFilterBiquad is
 * a fixed-point routine of the Audio library (not in the tree, not in the image's data): it stays build-defined
 * as this same float cascade with the int16 pack at its output.
 *
 * Mapping: the recursion is serial in time but the four stages of a cascade pipeline: lane s of a
 * quad runs stage s on sample n = i - s at step i and takes its input from lane s-1 (one DPP
 * quad_perm move), so a step costs one biquad instead of four in series and a wave carries
 * 16 channels.  Samples go through an LDS tile (coalesced 16-byte global loads and stores).
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "rdsp_host.h"
#include "rdsp_kernels.h"
#include "rdsp_sync.h"

namespace {

constexpr int TS = 128; /* samples per LDS tile */
constexpr int TP = TS + 4;
typedef float bq_v4 __attribute__((ext_vector_type(4)));

/* one sample of one section: products rounded, sum left to right.  The pragma is what keeps the compiler from
 * contracting product and sum into v_fmac_f32 (hipcc's default is -ffp-contract=fast, and __fmul_rn / __fadd_rn are
 * plain operators to it); tests/test_host_logic.py looks at the generated code */
__device__ __forceinline__ float df1_acc(float b0, float b1, float b2, float a1, float a2, float x, float x1, float x2,
                                         float y1, float y2) {
#pragma clang fp contract(off)
  float acc = b0 * x;
  acc = acc + b1 * x1;
  acc = acc + b2 * x2;
  acc = acc + a1 * y1;
  acc = acc + a2 * y2;
  return acc;
}

__device__ __forceinline__ int q15_trunc(float x) { /* arm_float_to_q15 (CONV:346-347, the image's rounding variant): x*32768, +-0.5 by sign, truncate, saturate */
#pragma clang fp contract(off) /* (x * 32768 is exact, so a fused form would give the same bits; kept unfused so that the kernel has no fused operation at all for the generated-code check to look for) */
  float v = x * 32768.0f;
  v = v + (v > 0.0f ? 0.5f : -0.5f);
  v = fminf(fmaxf(v, -32768.0f), 32767.0f);
  return (int)v;
}

__global__ void __launch_bounds__(64) rdsp_biquad_kernel(RdspBiquadParams p) {
  __shared__ __attribute__((aligned(16))) float tile[16][TP];
  const int lane = threadIdx.x;
  const int cl = lane >> 2, s = lane & 3;
  const int ch0 = p.ch_base + (int)blockIdx.x * 16;
  int ch = ch0 + cl;
  const bool valid = ch < p.n_channels;
  if (!valid) ch = p.n_channels - 1;
  const uint32_t set = p.set_of ? (uint32_t)p.set_of[ch] : 0u;
  const float *cf = p.coef + (size_t)set * 20 + 5 * s;
  const float b0 = cf[0], b1 = cf[1], b2 = cf[2], a1 = cf[3], a2 = cf[4];
  float *st = p.state + (size_t)ch * 16 + 4 * s;
  float x1 = st[0], x2 = st[1], y1 = st[2], y2 = st[3];
  /* float tiles: this lane's piece of rows (lane >> 5) + 2 it (rows past the last channel read a real one) */
  size_t row_off[8];
#pragma unroll
  for (int it = 0; it < 8; it++) {
    int rch = ch0 + (lane >> 5) + 2 * it;
    if (rch >= p.n_channels) rch = p.n_channels - 1;
    row_off[it] = (size_t)rch * p.stride + 4 * (lane & 31);
  }

#pragma unroll 1
  for (int t0 = 0; t0 < p.n_samples; t0 += TS) {
    /* tile in: 16 rows x 128 samples */
    if (p.in16) {
      for (int idx = lane; idx < 16 * (TS / 2); idx += 64) {
        const int row = idx / (TS / 2), c2 = idx % (TS / 2);
        int rch = ch0 + row;
        if (rch >= p.n_channels) rch = p.n_channels - 1;
        const int16_t *src = p.in16 + ((size_t)rch * p.stride16 + (size_t)(t0 + 2 * c2)) * p.step16;
        tile[row][2 * c2] = (float)src[0] * (1.0f / 32768.0f);
        tile[row][2 * c2 + 1] = (float)src[p.step16] * (1.0f / 32768.0f);
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; it++) /* 32 lanes per row, two rows per pass */
        *reinterpret_cast<bq_v4 *>(&tile[(lane >> 5) + 2 * it][4 * (lane & 31)]) =
            *reinterpret_cast<const bq_v4 *>(p.buf + row_off[it] + t0);
    }
    wg_sync<1>();
    /* skewed steps i = 0 .. TS + 2, lane s on sample n = i - s.  Steps 7 .. TS - 2 have every lane
     * inside the tile: they run unmasked, four to a chunk (i = 4c + 3 .. 4c + 6, c = 1 .. TS/4 - 2), so
     * that stage 3 finishes samples 4c .. 4c + 3 and stage 0 starts 4c + 3 .. 4c + 6 -- one 16-byte
     * LDS read and one write per chunk; the ramps at both ends of the tile keep the general form */
    float yprev = 0.f;
    auto ramp_step = [&](int i) {
      const int n = i - s;
      const bool active = (n >= 0) && (n < TS);
      const float xin0 = tile[cl][i < TS ? i : TS - 1];
      /* lane s takes lane s-1's output of the previous step: quad_perm [0,0,1,2] */
      const float xup = __builtin_bit_cast(
          float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, yprev), 0x90, 0xF, 0xF, false));
      const float x = (s == 0) ? xin0 : xup;
      const float y = df1_acc(b0, b1, b2, a1, a2, x, x1, x2, y1, y2);
      if (active) {
        x2 = x1; x1 = x;
        y2 = y1; y1 = y;
        yprev = y;
        if (s == 3) tile[cl][n] = y;
      }
    };
#pragma unroll
    for (int i = 0; i < 7; i++) ramp_step(i);
    {
      bq_v4 q = *reinterpret_cast<const bq_v4 *>(&tile[cl][4]);
#pragma unroll 2
      for (int c = 1; c <= TS / 4 - 2; c++) {
        const bq_v4 qn = *reinterpret_cast<const bq_v4 *>(&tile[cl][4 * c + 4]);
        const float xin[4] = {q[3], qn[0], qn[1], qn[2]};
        bq_v4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const float xup = __builtin_bit_cast(
              float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, yprev), 0x90, 0xF, 0xF, false));
          const float x = (s == 0) ? xin[k] : xup;
          const float y = df1_acc(b0, b1, b2, a1, a2, x, x1, x2, y1, y2);
          x2 = x1; x1 = x;
          y2 = y1; y1 = y;
          yprev = y;
          o[k] = y;
        }
        if (s == 3) *reinterpret_cast<bq_v4 *>(&tile[cl][4 * c]) = o;
        q = qn;
      }
    }
#pragma unroll
    for (int i = TS - 1; i < TS + 3; i++) ramp_step(i);
    wg_sync<1>();
    /* tile out */
    if (p.out16) {
      for (int idx = lane; idx < 16 * (TS / 2); idx += 64) {
        const int row = idx / (TS / 2), c2 = idx % (TS / 2);
        const int rch = ch0 + row;
        if (rch < p.n_channels) {
          int16_t *dst = p.out16 + ((size_t)rch * p.ostride16 + (size_t)(t0 + 2 * c2)) * p.ostep16;
          dst[0] = (int16_t)q15_trunc(tile[row][2 * c2]);
          dst[p.ostep16] = (int16_t)q15_trunc(tile[row][2 * c2 + 1]);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < 8; it++)
        if (ch0 + (lane >> 5) + 2 * it < p.n_channels)
          *reinterpret_cast<bq_v4 *>(p.buf + row_off[it] + t0) =
              *reinterpret_cast<const bq_v4 *>(&tile[(lane >> 5) + 2 * it][4 * (lane & 31)]);
    }
    wg_sync<1>();
  }
  if (valid) { st[0] = x1; st[1] = x2; st[2] = y1; st[3] = y2; }
}

struct CoefWords { float w[20]; };
__global__ void rdsp_biquad_coef_store_kernel(float *dst, CoefWords v) {
  if (threadIdx.x < 20) dst[threadIdx.x] = v.w[threadIdx.x];
}

}  // namespace

/* n_samples must be a multiple of 128 (one audio block) */
extern "C" int rdsp_launch_biquad(const RdspBiquadParams *p, hipStream_t stream) {
  if (p->n_samples <= 0 || p->n_samples % TS != 0 || p->n_channels <= p->ch_base) return (int)hipErrorInvalidValue;
  const int grid = (p->n_channels - p->ch_base + 15) / 16;
  hipLaunchKernelGGL(rdsp_biquad_kernel, dim3(grid), dim3(64), 0, stream, *p);
  return (int)hipGetLastError();
}
/* one coefficient set (20 floats) rewritten in stream order, values passed by value */
extern "C" int rdsp_launch_biquad_coef_store(float *dst, const float *coef20, hipStream_t stream) {
  CoefWords v;
  memcpy(v.w, coef20, sizeof(v.w));
  hipLaunchKernelGGL(rdsp_biquad_coef_store_kernel, dim3(1), dim3(32), 0, stream, dst, v);
  return (int)hipGetLastError();
}

/* ---- AudioFilterBiquad object: n_channels independent cascades with common coefficients ---- */
struct rdsp_biquad {
  int n_channels, device;
  double fs;
  float coef[20];
  bool dirty = true;
  float *d_coef = nullptr, *d_state = nullptr;
};

#define BQ_TRY(expr)                                                            \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      rdsp_set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
      return RDSP_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)

extern "C" int rdsp_biquad_create(int n_channels, int device, double fs, rdsp_biquad_t **out) {
  if (!out || n_channels <= 0 || !(fs > 0.0)) {
    rdsp_set_error("rdsp_biquad_create: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (rdsp_device_count() <= 0) {
    rdsp_set_error("no HIP device: the rdsp product path has no CPU fallback");
    return RDSP_ERR_NO_DEVICE;
  }
  rdsp_biquad_t *b = new rdsp_biquad();
  b->n_channels = n_channels;
  b->device = device;
  b->fs = fs;
  for (int s = 0; s < 4; s++) { /* a fresh AudioFilterBiquad passes audio through */
    b->coef[5 * s] = 1.0f;
    b->coef[5 * s + 1] = b->coef[5 * s + 2] = b->coef[5 * s + 3] = b->coef[5 * s + 4] = 0.0f;
  }
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&b->d_coef, sizeof(b->coef)) != hipSuccess ||
      hipMalloc((void **)&b->d_state, sizeof(float) * 16 * (size_t)n_channels) != hipSuccess ||
      hipMemset(b->d_state, 0, sizeof(float) * 16 * (size_t)n_channels) != hipSuccess) {
    rdsp_set_error("rdsp_biquad_create: device allocation failed");
    rdsp_biquad_destroy(b);
    return RDSP_ERR_HIP;
  }
  *out = b;
  return RDSP_OK;
}
extern "C" void rdsp_biquad_destroy(rdsp_biquad_t *b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  if (b->d_coef) (void)hipFree(b->d_coef);
  if (b->d_state) (void)hipFree(b->d_state);
  delete b;
}
/* AudioFilterBiquad::setCoefficients(stage, const double *): {b0, b1, b2, a1, a2} of
 * H(z) = (b0 + b1 z^-1 + b2 z^-2) / (1 + a1 z^-1 + a2 z^-2) */
extern "C" int rdsp_biquad_setCoefficients(rdsp_biquad_t *b, int stage, const double *c5) {
  if (!b || !c5 || stage < 0 || stage > 3) return RDSP_ERR_INVALID;
  float *c = b->coef + 5 * stage;
  c[0] = (float)c5[0]; c[1] = (float)c5[1]; c[2] = (float)c5[2];
  c[3] = (float)(-c5[3]); c[4] = (float)(-c5[4]);
  b->dirty = true;
  return RDSP_OK;
}
static int set_design(rdsp_biquad_t *b, int stage, int kind, float freq, float q) {
  if (!b || stage < 0 || stage > 3 || !(freq > 0.f) || !(q > 0.f)) return RDSP_ERR_INVALID;
  rdsp_biquad_design(kind, (double)freq, (double)q, b->fs, b->coef + 5 * stage);
  b->dirty = true;
  return RDSP_OK;
}
extern "C" int rdsp_biquad_setLowpass(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 0, f, q); }
extern "C" int rdsp_biquad_setHighpass(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 1, f, q); }
extern "C" int rdsp_biquad_setBandpass(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 2, f, q); }
extern "C" int rdsp_biquad_setNotch(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 3, f, q); }
extern "C" int rdsp_biquad_get_coeffs(const rdsp_biquad_t *b, float *out20) {
  if (!b || !out20) return RDSP_ERR_INVALID;
  memcpy(out20, b->coef, sizeof(b->coef));
  return RDSP_OK;
}

/* n_blocks update() ticks for every channel: int16 audio in, int16 audio out.
 * d_in / d_out: [n_channels][stride] samples taken / written every `step` int16 (1: planar mono
 * blocks, 2: one side of interleaved pairs, e.g. I or Q of an IQ stream). */
extern "C" int rdsp_biquad_update(rdsp_biquad_t *b, const int16_t *d_in, size_t in_stride, int in_step, int n_blocks,
                                  int16_t *d_out, size_t out_stride, int out_step, void *stream_) {
  if (!b || !d_in || !d_out || n_blocks <= 0 || in_step < 1 || out_step < 1 ||
      in_stride < (size_t)n_blocks * 128 || out_stride < (size_t)n_blocks * 128) {
    rdsp_set_error("rdsp_biquad_update: bad argument");
    return RDSP_ERR_INVALID;
  }
  BQ_TRY(hipSetDevice(b->device));
  hipStream_t stream = (hipStream_t)stream_;
  if (b->dirty) {
    int e = rdsp_launch_biquad_coef_store(b->d_coef, b->coef, stream);
    if (e != 0) { rdsp_set_error("biquad coefficient store: %s", hipGetErrorString((hipError_t)e)); return RDSP_ERR_HIP; }
    b->dirty = false;
  }
  RdspBiquadParams p;
  memset(&p, 0, sizeof(p));
  p.in16 = d_in; p.stride16 = in_stride; p.step16 = in_step;
  p.out16 = d_out; p.ostride16 = out_stride; p.ostep16 = out_step;
  p.n_channels = b->n_channels;
  p.n_samples = n_blocks * 128;
  p.coef = b->d_coef;
  p.state = b->d_state;
  int e = rdsp_launch_biquad(&p, stream);
  if (e != 0) { rdsp_set_error("biquad kernel launch failed: %s", hipGetErrorString((hipError_t)e)); return RDSP_ERR_HIP; }
  return RDSP_OK;
}

/* ---- the node: `AudioFilterBiquad biquad1;` (INO:58), one input, one output ------------------- */
namespace {
struct BiquadNode {
  rdsp_biquad_t *bq;
  int n_channels;
  int16_t *d_in = nullptr, *d_out = nullptr;
  hipStream_t stream = nullptr;
  int status = RDSP_OK;
  int device = 0; /* the object's device: selected in update and destroy (a process may drive several GPUs) */
};
void biquad_node_destroy(void *u) {
  BiquadNode *s = static_cast<BiquadNode *>(u);
  (void)hipSetDevice(s->device); /* the node's buffers and stream live on its object's device */
  if (s->d_in) (void)hipFree(s->d_in);
  if (s->d_out) (void)hipFree(s->d_out);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}
void biquad_node_update(rdsp_node_t *n, void *u) {
  BiquadNode *s = static_cast<BiquadNode *>(u);
  rdsp_block_t *in = rdsp_receive_readonly(n, 0);
  if (!in) return; /* no input this tick: nothing is transmitted */
  (void)hipSetDevice(s->device);
  rdsp_block_t *out = rdsp_allocate(n);
  if (!out) { rdsp_release(in); return; }
  const size_t bytes = (size_t)s->n_channels * RDSP_BLOCK_SAMPLES * sizeof(int16_t);
  hipError_t e = hipMemcpyAsync(s->d_in, rdsp_block_data(in), bytes, hipMemcpyHostToDevice, s->stream);
  int rc = RDSP_OK;
  if (e == hipSuccess) rc = rdsp_biquad_update(s->bq, s->d_in, RDSP_BLOCK_SAMPLES, 1, 1, s->d_out, RDSP_BLOCK_SAMPLES, 1, s->stream);
  if (e == hipSuccess && rc == RDSP_OK) e = hipMemcpyAsync(rdsp_block_data(out), s->d_out, bytes, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess && rc == RDSP_OK) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess || rc != RDSP_OK) {
    s->status = (rc != RDSP_OK) ? rc : RDSP_ERR_HIP;
    if (e != hipSuccess) rdsp_set_error("biquad node: %s", hipGetErrorString(e));
  } else {
    rdsp_transmit(n, out, 0);
  }
  rdsp_release(out);
  rdsp_release(in);
}
}  // namespace

extern "C" rdsp_node_t *rdsp_biquad_node_create(rdsp_graph_t *g, rdsp_biquad_t *bq) {
  if (!g || !bq || bq->n_channels != rdsp_graph_channels(g)) {
    rdsp_set_error("rdsp_biquad_node_create: bad argument (the cascade needs the graph's channel count)");
    return nullptr;
  }
  BiquadNode *s = new BiquadNode();
  s->bq = bq;
  s->n_channels = bq->n_channels;
  s->device = bq->device;
  const size_t bytes = (size_t)s->n_channels * RDSP_BLOCK_SAMPLES * sizeof(int16_t);
  if (hipSetDevice(bq->device) != hipSuccess || hipMalloc((void **)&s->d_in, bytes) != hipSuccess ||
      hipMalloc((void **)&s->d_out, bytes) != hipSuccess || hipStreamCreate(&s->stream) != hipSuccess) {
    rdsp_set_error("rdsp_biquad_node_create: device allocation failed");
    biquad_node_destroy(s);
    return nullptr;
  }
  rdsp_node_t *n = rdsp_node_create(g, 1, biquad_node_update, s);
  if (!n) { biquad_node_destroy(s); return nullptr; }
  rdsp_node_set_destructor(n, biquad_node_destroy);
  return n;
}
extern "C" int rdsp_biquad_node_status(rdsp_node_t *n) {
  BiquadNode *s = static_cast<BiquadNode *>(rdsp_node_user(n));
  return s ? s->status : RDSP_ERR_INVALID;
}
