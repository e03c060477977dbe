/*
 * rdsp_biquad.hip -- biquad cascades on the GPU.
 *
 * Two callers, two routines of two libraries (neither in the tree), one kernel body with two section types:
 *   * the engine's IIR audio filter bank: SDR.setAudioFilter(audioCW ... audioAM)
 *     (RDSP_controls.h:153-177) is a table of 8th-order band-passes in
 *     arm_biquad_cascade_df1_f32 layout in the shipped firmware; the chain
 *     runs it on the demodulated mono audio between the front kernel and the tail stage when
 *     rdsp_sdr_setAudioFilterKind selects it (rdsp_chain.hip).  SectionF32: arm_biquad_cascade_df1_f32 as
 *     CMSIS-DSP publishes it, direct form 1 in float, per stage
 *         acc = (b0 * Xn) + (b1 * Xn1) + (b2 * Xn2) + (a1 * Yn1) + (a2 * Yn2)
 *     summed left to right, EVERY PRODUCT ROUNDED BEFORE IT IS ADDED (feedback terms added, coefficient order
 *     {b0, b1, b2, a1, a2}) -- the routine is in the reference's firmware image as 5 VMUL + 4 VADD per sample, no
 *     fused operation (tests/test_firmware_tables.py); until round 5 this was a chain of four fused multiply-adds.
 *     Four stages, unused ones pass through exactly.
 *   * AudioFilterBiquad nodes, `biquad1.setHighpass(0, 500, 0.5)` in front of the panadapter
 *     (RadioDSP_SDR_RX.ino:58-59,75-78,155-156): rdsp_biquad_t + rdsp_biquad_node_create.  SectionTeensy: the Teensy
 *     Audio library's FIXED-POINT update() as the firmware image holds it (0xe1b8 ... 0xe21e: SMLAWB / SMLAWT x 5,
 *     SSAT #16 ASR #14, UBFX #0 #14, twice per loop turn, PKHBT): int32 coefficients x 2^30 with a1, a2 stored
 *     negated, per sample
 *         sum += (b0 x) >> 16; sum += (b1 x1) >> 16; sum += (b2 x2) >> 16; sum += (a1 y1) >> 16; sum += (a2 y2) >> 16
 *         y = ssat16(sum >> 14); sum &= 0x3FFF
 *     on 48-bit products and a 32-bit wrap-around accumulator; history is int16 (y1, y2 are the SATURATED outputs);
 *     update() runs stage 0 and goes on to stage s + 1 only if setCoefficients(s + 1) was ever called; a fresh object
 *     has all-zero coefficients and passes nothing.  Until round 5 this object was the float cascade above with the
 *     int16 pack at its output -- up to a count away from what the reference's hardware computes, and passing the
 *     signal through where the reference passes nothing.
 *
 * Mapping: the recursion is serial in time but the four stages of a cascade pipeline: lane s of a
 * quad runs stage s on sample n = i - s at step i and takes its input from lane s-1 (one DPP
 * quad_perm move), so a step costs one biquad instead of four in series and a wave carries
 * 16 channels.  Samples go through an LDS tile (coalesced 16-byte global loads and stores).
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include <type_traits>
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

/* one section of a cascade in lane s of a quad: the two arithmetics the kernel runs.  W is the 32-bit word that sits
 * in the LDS tile and travels between the stages by DPP. */
struct SectionF32 { /* arm_biquad_cascade_df1_f32 (the chain's IIR bank; float audio in place) */
  float b0, b1, b2, a1, a2, x1, x2, y1, y2;
  float *st;
  __device__ __forceinline__ void load(const RdspBiquadParams &p, int ch, int s) {
    const uint32_t set = p.set_of ? (uint32_t)p.set_of[ch] : 0u;
    const float *cf = p.coef + (size_t)set * 20 + 5 * s;
    b0 = cf[0]; b1 = cf[1]; b2 = cf[2]; a1 = cf[3]; a2 = cf[4];
    st = p.state + (size_t)ch * 16 + 4 * s;
    x1 = st[0]; x2 = st[1]; y1 = st[2]; y2 = st[3];
  }
  __device__ __forceinline__ float eval(float x) const { return df1_acc(b0, b1, b2, a1, a2, x, x1, x2, y1, y2); }
  __device__ __forceinline__ void commit(float x, float y) { x2 = x1; x1 = x; y2 = y1; y1 = y; }
  __device__ __forceinline__ void store() const { st[0] = x1; st[1] = x2; st[2] = y1; st[3] = y2; }
};
/* AudioFilterBiquad::update of the Teensy Audio library (filter_biquad.cpp), per sample: five 32 x 16 products that keep
 * the top 32 of 48 bits (SMLAWB / SMLAWT) on top of the 14 fractional bits of the previous sample's sum, output
 * signed_saturate_rshift(sum, 16, 14), `sum &= 0x3FFF` stays behind.  W = the int16 sample, sign-extended. */
struct SectionTeensy {
  int c0, c1, c2, c3, c4, x1, x2, y1, y2, sum;
  int *st;
  bool on; /* stages behind the cascade's last one do not run: the audio passes them untouched */
  static __device__ __forceinline__ int smlaw(int acc, int c, int v) { /* SMLAW wraps: the sum is formed in unsigned arithmetic */
    return (int)((unsigned)acc + (unsigned)(int)(((long long)c * (long long)v) >> 16));
  }
  __device__ __forceinline__ void load(const RdspBiquadParams &p, int ch, int s) {
    const int *cf = p.icoef + 5 * s;
    c0 = cf[0]; c1 = cf[1]; c2 = cf[2]; c3 = cf[3]; c4 = cf[4];
    on = s < (p.n_stages > 0 ? p.n_stages : 1); /* update()'s do-while runs stage 0 even when nothing was ever set */
    st = p.istate + (size_t)ch * 20 + 5 * s;
    x1 = st[0]; x2 = st[1]; y1 = st[2]; y2 = st[3]; sum = st[4];
  }
  __device__ __forceinline__ int eval(int x) {
    if (!on) return x;
    int acc = sum & 0x3FFF;
    acc = smlaw(acc, c0, x); acc = smlaw(acc, c1, x1); acc = smlaw(acc, c2, x2); acc = smlaw(acc, c3, y1); acc = smlaw(acc, c4, y2);
    pending = acc;
    const int y = acc >> 14;
    return y > 32767 ? 32767 : (y < -32768 ? -32768 : y);
  }
  __device__ __forceinline__ void commit(int x, int y) { if (on) { x2 = x1; x1 = x; y2 = y1; y1 = y; sum = pending & 0x3FFF; } }
  __device__ __forceinline__ void store() const { st[0] = x1; st[1] = x2; st[2] = y1; st[3] = y2; st[4] = sum; }
  int pending;
};

template <typename SEC, typename W>
__device__ __forceinline__ void biquad_body(const RdspBiquadParams &p, W (*tile)[TP]) {
  const int lane = threadIdx.x;
  const int cl = lane >> 2, s = lane & 3;
  const int ch0 = p.ch_base + (int)blockIdx.x * 16;
  int ch = ch0 + cl;
  const bool valid = ch < p.n_channels;
  if (!valid) ch = p.n_channels - 1;
  SEC sec;
  sec.load(p, ch, s);
  constexpr bool FIXED = std::is_same<SEC, SectionTeensy>::value;
  /* float tiles: this lane's piece of rows (lane >> 5) + 2 it (rows past the last channel read a real one) */
  size_t row_off[8];
#pragma unroll
  for (int it = 0; it < 8; it++) {
    int rch = ch0 + (lane >> 5) + 2 * it;
    if (rch >= p.n_channels) rch = p.n_channels - 1;
    row_off[it] = (size_t)rch * p.stride + 4 * (lane & 31);
  }
  auto dpp_up = [](W v) { /* lane s takes lane s-1's value: quad_perm [0,0,1,2] */
    return __builtin_bit_cast(W, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x90, 0xF, 0xF, false));
  };

#pragma unroll 1
  for (int t0 = 0; t0 < p.n_samples; t0 += TS) {
    /* tile in: 16 rows x 128 samples */
    if (p.in16) {
      for (int idx = lane; idx < 16 * (TS / 2); idx += 64) {
        const int row = idx / (TS / 2), c2 = idx % (TS / 2);
        int rch = ch0 + row;
        if (rch >= p.n_channels) rch = p.n_channels - 1;
        const int16_t *src = p.in16 + ((size_t)rch * p.stride16 + (size_t)(t0 + 2 * c2)) * p.step16;
        if constexpr (FIXED) {
          tile[row][2 * c2] = (W)src[0];
          tile[row][2 * c2 + 1] = (W)src[p.step16];
        } else {
          tile[row][2 * c2] = (W)((float)src[0] * (1.0f / 32768.0f));
          tile[row][2 * c2 + 1] = (W)((float)src[p.step16] * (1.0f / 32768.0f));
        }
      }
    } else if constexpr (!FIXED) {
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
    W yprev = (W)0;
    auto ramp_step = [&](int i) {
      const int n = i - s;
      const bool active = (n >= 0) && (n < TS);
      const W xin0 = tile[cl][i < TS ? i : TS - 1];
      const W xup = dpp_up(yprev);
      const W x = (s == 0) ? xin0 : xup;
      const W y = sec.eval(x);
      if (active) {
        sec.commit(x, y);
        yprev = y;
        if (s == 3) tile[cl][n] = y;
      }
    };
#pragma unroll
    for (int i = 0; i < 7; i++) ramp_step(i);
    {
      typedef W w4 __attribute__((ext_vector_type(4)));
      w4 q = *reinterpret_cast<const w4 *>(&tile[cl][4]);
#pragma unroll 2
      for (int c = 1; c <= TS / 4 - 2; c++) {
        const w4 qn = *reinterpret_cast<const w4 *>(&tile[cl][4 * c + 4]);
        const W xin[4] = {q[3], qn[0], qn[1], qn[2]};
        w4 o;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const W xup = dpp_up(yprev);
          const W x = (s == 0) ? xin[k] : xup;
          const W y = sec.eval(x);
          sec.commit(x, y);
          yprev = y;
          o[k] = y;
        }
        if (s == 3) *reinterpret_cast<w4 *>(&tile[cl][4 * c]) = o;
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
          if constexpr (FIXED) {
            dst[0] = (int16_t)tile[row][2 * c2];
            dst[p.ostep16] = (int16_t)tile[row][2 * c2 + 1];
          } else {
            dst[0] = (int16_t)q15_trunc((float)tile[row][2 * c2]);
            dst[p.ostep16] = (int16_t)q15_trunc((float)tile[row][2 * c2 + 1]);
          }
        }
      }
    } else if constexpr (!FIXED) {
#pragma unroll
      for (int it = 0; it < 8; it++)
        if (ch0 + (lane >> 5) + 2 * it < p.n_channels)
          *reinterpret_cast<bq_v4 *>(p.buf + row_off[it] + t0) =
              *reinterpret_cast<const bq_v4 *>(&tile[(lane >> 5) + 2 * it][4 * (lane & 31)]);
    }
    wg_sync<1>();
  }
  if (valid) sec.store();
}

__global__ void __launch_bounds__(64) rdsp_biquad_kernel(RdspBiquadParams p) {
  __shared__ __attribute__((aligned(16))) float tile[16][TP];
  biquad_body<SectionF32, float>(p, tile);
}
/* AudioFilterBiquad objects: int16 in, int16 out, the library's fixed-point arithmetic */
__global__ void __launch_bounds__(64) rdsp_biquad_teensy_kernel(RdspBiquadParams p) {
  __shared__ __attribute__((aligned(16))) int tile[16][TP];
  biquad_body<SectionTeensy, int>(p, tile);
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
  if (p->icoef) hipLaunchKernelGGL(rdsp_biquad_teensy_kernel, dim3(grid), dim3(64), 0, stream, *p);
  else hipLaunchKernelGGL(rdsp_biquad_kernel, dim3(grid), dim3(64), 0, stream, *p);
  return (int)hipGetLastError();
}
/* one coefficient set (20 floats) rewritten in stream order, values passed by value */
extern "C" int rdsp_launch_biquad_coef_store(float *dst, const float *coef20, hipStream_t stream) {
  CoefWords v;
  memcpy(v.w, coef20, sizeof(v.w));
  hipLaunchKernelGGL(rdsp_biquad_coef_store_kernel, dim3(1), dim3(32), 0, stream, dst, v);
  return (int)hipGetLastError();
}

/* ---- AudioFilterBiquad object: n_channels independent cascades with common coefficients ----------------
 * The Teensy Audio library's class (filter_biquad.{h,cpp}; not in the reference tree) restated from its published
 * source: `int32_t definition[32]` = four stages of {b0, b1, b2, -a1, -a2} scaled by 2^30 and three words of state.
 * The image of the reference confirms update()'s operation sequence and the setters' float constant 2 pi / 44100
 * (tests/test_firmware_tables.py). */
struct rdsp_biquad {
  int n_channels, device;
  double fs;
  int32_t coef[4][5]; /* as definition[] holds them: a1, a2 negated */
  bool chained[4] = {false, false, false, false}; /* `if (stage > 0) *(dest - 1) |= 0x80000000`: stage - 1 hands on to stage */
  int n_stages = 0;   /* update() runs stage 0, then every next stage while the one before it is chained to it */
  bool dirty = true;
  std::vector<int> clear_sum; /* stages whose residue the next update zeroes (`*dest &= 0x80000000`) */
  int32_t *d_coef = nullptr, *d_state = nullptr;
};

#define BQ_TRY(expr)                                                            \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      rdsp_set_error("%s failed: %s", #expr, hipGetErrorString(e_));            \
      return RDSP_ERR_HIP;                                                      \
    }                                                                           \
  } while (0)

namespace {
struct ICoefWords { int32_t w[20]; };
__global__ void rdsp_biquad_icoef_store_kernel(int32_t *dst, ICoefWords v) {
  if (threadIdx.x < 20) dst[threadIdx.x] = v.w[threadIdx.x];
}
/* the residue word of one stage of every channel */
__global__ void rdsp_biquad_clear_sum_kernel(int32_t *state, int n_channels, int stage) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < n_channels) state[(size_t)ch * 20 + 5 * stage + 4] = 0;
}
}  // namespace

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
  memset(b->coef, 0, sizeof(b->coef)); /* "by default, the filter will not pass anything" (the library's constructor) */
  if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&b->d_coef, sizeof(b->coef)) != hipSuccess ||
      hipMalloc((void **)&b->d_state, sizeof(int32_t) * 20 * (size_t)n_channels) != hipSuccess ||
      hipMemset(b->d_state, 0, sizeof(int32_t) * 20 * (size_t)n_channels) != hipSuccess) {
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
/* void setCoefficients(uint32_t stage, const int *coefficients): {b0, b1, b2, a1, a2} x 2^30; a1 and a2 are stored
 * negated, the stage's residue is cleared, its sample history stays ("clearing filter state causes loud pop") */
extern "C" int rdsp_biquad_setCoefficients_int(rdsp_biquad_t *b, int stage, const int32_t *c5) {
  if (!b || !c5 || stage < 0 || stage > 3) return RDSP_ERR_INVALID;
  int32_t *c = b->coef[stage];
  c[0] = c5[0]; c[1] = c5[1]; c[2] = c5[2];
  c[3] = (int32_t)(0u - (uint32_t)c5[3]);
  c[4] = (int32_t)(0u - (uint32_t)c5[4]);
  if (stage > 0) b->chained[stage - 1] = true;
  b->n_stages = 1;
  while (b->n_stages < 4 && b->chained[b->n_stages - 1]) b->n_stages++;
  b->clear_sum.push_back(stage);
  b->dirty = true;
  return RDSP_OK;
}
/* void setCoefficients(uint32_t stage, const double *coefficients) of
 * H(z) = (b0 + b1 z^-1 + b2 z^-2) / (1 + a1 z^-1 + a2 z^-2): each times 1073741824.0, converted to int */
extern "C" int rdsp_biquad_setCoefficients(rdsp_biquad_t *b, int stage, const double *c5) {
  if (!b || !c5 || stage < 0 || stage > 3) return RDSP_ERR_INVALID;
  int32_t ci[5];
  for (int i = 0; i < 5; i++) {
    const double v = c5[i] * 1073741824.0;
    if (!(v > -2147483649.0 && v < 2147483648.0)) {
      rdsp_set_error("rdsp_biquad_setCoefficients: coefficient %d = %g does not fit the library's 2.30 format", i, c5[i]);
      return RDSP_ERR_INVALID;
    }
    ci[i] = (int32_t)v;
  }
  return rdsp_biquad_setCoefficients_int(b, stage, ci);
}
static int set_design(rdsp_biquad_t *b, int stage, int kind, float freq, float q) {
  if (!b || stage < 0 || stage > 3 || !(freq > 0.f) || !(q > 0.f) || !((double)freq < 0.5 * b->fs)) return RDSP_ERR_INVALID; /* at or past fs / 2 the cookbook's x 2^30 coefficients leave the int32 range */
  int32_t ci[5];
  rdsp_teensy_biquad_design(kind, freq, q, (float)b->fs, ci);
  return rdsp_biquad_setCoefficients_int(b, stage, ci);
}
extern "C" int rdsp_biquad_setLowpass(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 0, f, q); }
extern "C" int rdsp_biquad_setHighpass(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 1, f, q); }
extern "C" int rdsp_biquad_setBandpass(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 2, f, q); }
extern "C" int rdsp_biquad_setNotch(rdsp_biquad_t *b, int stage, float f, float q) { return set_design(b, stage, 3, f, q); }
/* the coefficient words as definition[] holds them, and how many stages the cascade runs */
extern "C" int rdsp_biquad_get_definition(const rdsp_biquad_t *b, int32_t *out20, int *n_stages) {
  if (!b || !out20) return RDSP_ERR_INVALID;
  memcpy(out20, b->coef, sizeof(b->coef));
  if (n_stages) *n_stages = b->n_stages;
  return RDSP_OK;
}
/* the same as floats, {b0, b1, b2, a1, a2} with the feedback terms added (coefficient / 2^30) */
extern "C" int rdsp_biquad_get_coeffs(const rdsp_biquad_t *b, float *out20) {
  if (!b || !out20) return RDSP_ERR_INVALID;
  for (int i = 0; i < 20; i++) out20[i] = (float)((double)b->coef[i / 5][i % 5] / 1073741824.0);
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
    ICoefWords v;
    memcpy(v.w, b->coef, sizeof(v.w));
    hipLaunchKernelGGL(rdsp_biquad_icoef_store_kernel, dim3(1), dim3(32), 0, stream, b->d_coef, v);
    for (int st : b->clear_sum)
      hipLaunchKernelGGL(rdsp_biquad_clear_sum_kernel, dim3((b->n_channels + 255) / 256), dim3(256), 0, stream, b->d_state,
                         b->n_channels, st);
    b->clear_sum.clear();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { rdsp_set_error("biquad coefficient store: %s", hipGetErrorString(e)); return RDSP_ERR_HIP; }
    b->dirty = false;
  }
  RdspBiquadParams p;
  memset(&p, 0, sizeof(p));
  p.in16 = d_in; p.stride16 = in_stride; p.step16 = in_step;
  p.out16 = d_out; p.ostride16 = out_stride; p.ostep16 = out_step;
  p.n_channels = b->n_channels;
  p.n_samples = n_blocks * 128;
  p.icoef = b->d_coef;
  p.n_stages = b->n_stages;
  p.istate = b->d_state;
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
