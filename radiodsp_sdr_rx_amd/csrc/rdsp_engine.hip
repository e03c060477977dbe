/*
 * rdsp_engine.hip -- `AudioSDR SDR;` (RadioDSP_SDR_RX.ino:54; wired :81-86, configured :117-139, driven from
 * RDSP_controls.h:149-423) for n_channels receivers on one GPU, computing what the reference's engine computes.
 *
 * The engine is Derek Rowell's AudioSDR library, which is not in the reference tree; what the reference holds of it is
 * its compiled code in pre_compiled/RadioDSP_SDR_RX.ino.hex (AudioSDR::update at ITCM 0xe730).  The arithmetic below
 * follows that code stage by stage -- which products are rounded before they are added, which are fused, where it widens
 * to double, its truncating conversions -- so that, on the same int16 IQ, this object returns the int16 audio the
 * image's update() returns (tests/test_engine_kat.py: the known answers are the image's own, recorded under the
 * interpreter of tests/golden/).  It is a low-IF receiver: IQ / 32767 x gains -> [impulse blanker] -> IF band-pass
 * (4 biquad sections per rail) -> table oscillator shifts the carrier to 0 Hz -> I delayed 128 samples, Q through a
 * 257-tap Hilbert transformer, sum or difference picks the side band  (AM: second IF pass, shift by the IF, low-pass,
 * envelope; SAM: a PLL on the IF signal with the envelope detector as its out-of-lock fallback) -> audio band-pass
 * -> hang AGC (peak envelope, gain by a curve) -> ALS line enhancer (delayed-input LMS) -> x output gain x 32767.
 *
 * Mapping to the machine.  Every stage but the Hilbert transformer is a recursion in time that has to be evaluated in the
 * image's order, so the parallel axes are channels, rails and -- for the transformer -- samples; and a lone wave issues an
 * instruction every five cycles or so whatever it depends on, so a launch lasts as long as ONE workgroup's chain of
 * instructions per block, times the blocks.  Hence: few channels per workgroup (8), the sections of a cascade on the four
 * lanes of a quad (sample n enters section s at step n + s), everything that is a pure function of a recursion's state
 * (table look-ups, gain curve, conversions, pack) spread over all lanes behind a recursion reduced to its few dependent
 * operations, and the passes of a block on different waves, each a block behind the previous one's:
 *   rdsp_engine_front_pipe_kernel  SSB / CW, no blanker: waves 2 + 3 convert block s and rotate / store block s - 2, wave 0
 *                              runs the IF cascades of block s - 1 (16 rows x 4 sections), wave 1 the oscillator's phase
 *   rdsp_engine_front_kernel   blanker, AM, SAM: the same passes one after the other (the PLL and the blanker's running
 *                              average are long recursions of their own)
 *   rdsp_engine_hilbert_kernel eight outputs of one parity per lane, sliding windows in registers, LDS split by parity
 *   rdsp_engine_tail_pipe_kernel   waves 2 + 3 load block s and finish block s - 3 (gain by the curve, clamp, pack), wave 0
 *                              the audio cascade of block s - 1, wave 1 the AGC envelope of block s - 2
 *   rdsp_engine_tail_kernel    with the ALS filter: its 55-tap chain on the lanes of a quad (the four samples between two
 *                              tap moves), taps in registers
 * int16 rows enter and leave in coalesced 256-byte segments through LDS tiles at a 129-word pitch.  The three launches of a
 * call are stream-ordered; a call takes any number of 128-sample blocks up to the engine's max_blocks_per_call.  HBM
 * traffic is 8 B per sample of algorithm (int16 IQ in, int16 L = R out) plus 28 B of float intermediates (the ring and the
 * audio row): the path is bound by the latency of its recursions, not by bandwidth.
 *
 * Three of the engine's tables have no closed form (fifteen sets of four biquad sections, 64 Hilbert taps): the host
 * loads them (rdsp_engine_load_tables; tests take them from tests/golden/firmware_tables.npz); update() refuses to run
 * without them.  The sine table and the AGC's gain curve are generated here the way the library generates them.
 *
 * Compiled with -ffp-contract=off: every fused operation below is written as one (fmaf / fma).
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rdsp_host.h"
#include "rdsp_sync.h"

namespace {

constexpr int BS = RDSP_BLOCK_SAMPLES;
constexpr int PITCH = BS + 1;
constexpr float TWO_PI_F = 6.2831854820251465f;   /* the float the image holds for 2 pi */
constexpr float RAD_PER_HZ = 0.00014247586659621447f; /* 2 pi / 44100, its float */

/* per-channel state, floats (ints bit-cast): [channel][NF] */
enum { ST_PRE = 0, ST_AM = 32, ST_AUDIO = 64, ST_NCO = 80, ST_AMPH, ST_SAM_COS, ST_SAM_SIN, ST_SAM_U, ST_SAM_ERR, ST_SAM_HZ,
       ST_SAM_PH, ST_SAM_LOCK, ST_AGC_ENV, ST_AGC_GAIN, ST_AGC_HANG, ST_AGC_ACTIVE, ST_NB_AVG, ST_NB_HIT, ST_NB_LAST, NF = 96 };
enum { RESET_PRE = 1, RESET_AUDIO = 2, RESET_ALS = 4 };
constexpr int ALS_TAPS = 55, ALS_DELAY = 3, ALS_WPITCH = 64; /* the constructor's values; the image has no setter for them */
constexpr int ALS_WORDS = 256 + ALS_WPITCH;                /* per channel in HBM: the 256-sample line, then the taps */
constexpr int NB_WORDS = 3 * 384;                          /* per channel: I line, Q line, mask */

struct EngParams {
  const int32_t *iq; size_t in_stride;   /* [ch][t] words: I | Q << 16 */
  int32_t *out; size_t out_stride;       /* [ch][t] words: L | R << 16 */
  int n_channels, n_blocks;
  float *st;
  float *ring_i, *ring_q; uint32_t ring_size, pos; /* [ch][ring_size], power of two; pos = where this call's first sample goes */
  float *audio; size_t audio_stride;     /* [ch][max samples per call] */
  float *nb, *als;
  const float *sets, *hilbert, *sine, *curve;
  int mode, mute, audio_on, agc_on, als_notch, als_adaptive, resets;
  int pre_set, audio_set;
  float gain_i, gain_q, output_gain, tuning_offset, if_centre;
  float agc_attack_a, agc_attack_b, agc_decay_a, agc_decay_b, agc_makeup; int agc_hang_time;
  float nb_keep, nb_new, nb_ratio; int nb_before, nb_after;
  float sam_keep, sam_new, sam_hz_per_rad, sam_lock_lo, sam_lock_hi, sam_ga, sam_gb;
};

__device__ __forceinline__ int trunc_s32(double x) { /* VCVT.S32.F64: toward zero, saturating, NaN -> 0 -- which is what v_cvt_i32_f64 does too */
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

/* the oscillator: sin of a phase in [0, 2 pi) by linear interpolation in the 256-step table, through double as the image does */
/* trunc(RN(a / d)) for a >= 0 and d = the double of the image's 2 pi, without the division: k d is exact for k < 2^16 (a
 * 24-bit d), so floor(a / d) follows from two exact comparisons around the estimate a (1 / d); and the correctly rounded
 * quotient cannot lie across an integer from the true one, because a is either exactly k d or at least an ulp of a away
 * from it, which is more than half an ulp of the quotient (tests/test_host_logic.py walks every k and its neighbours) */
__device__ __forceinline__ int index_of_phase(double a) {
  const double d = (double)TWO_PI_F;
  int k = (int)(a * (1.0 / d));
  if ((double)k * d > a) k--;
  else if ((double)(k + 1) * d <= a) k++;
  return k;
}
__device__ __forceinline__ float table_sin(const float *sine, float ph) {
  const int idx = index_of_phase((double)ph * 65535.0);
  const int hi = (idx >> 8) & 0xff;
  const float lo = (float)(unsigned)(idx & 0xff);
  const float t0 = sine[hi], t1 = sine[hi + 1];
  return (float)fma((double)((t1 - t0) * lo), 0.00390625, (double)t0);
}
__device__ __forceinline__ float dpp_up1(float v) { /* lane s of a quad takes lane s - 1's value: quad_perm [0,0,1,2] */
  const int w = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(w, w, 0x90, 0xF, 0xF, false)); /* every lane has a source: `old` is never kept */
}
__device__ __forceinline__ float quick_root_guess(float p) { return __uint_as_float((__float_as_uint(p) >> 1) + 0x1fa00000u + 0x1b4000u + 3886u); }
__device__ __forceinline__ float quick_sqrt1(float p) { const float g = quick_root_guess(p); return (p / g + g) * 0.5f; }
__device__ __forceinline__ float quick_sqrt2(float p) { const float y = quick_sqrt1(p); return (p / y + y) * 0.5f; }

/* ---- stages of a cascade on neighbouring lanes --------------------------------------------------------------------
 * arm_biquad_cascade_df1_f32 runs section after section over the block; the result is the same when sample n enters
 * section s at step n + s.  Lane s of a quad holds section s of one row of the tile (its five coefficients and four
 * state words) and at step i works on sample i - s, taking its input from lane s - 1's previous output (one DPP move):
 * a block costs 131 steps of ONE section instead of 128 of four, and a row occupies four lanes. */
struct Section {
  float b0, b1, b2, a1, a2, x1, x2, y1, y2;
  __device__ __forceinline__ void load(const float *coef5, const float *state4, bool clear) {
    b0 = coef5[0]; b1 = coef5[1]; b2 = coef5[2]; a1 = coef5[3]; a2 = coef5[4];
    x1 = clear ? 0.0f : state4[0]; x2 = clear ? 0.0f : state4[1]; y1 = clear ? 0.0f : state4[2]; y2 = clear ? 0.0f : state4[3];
  }
  __device__ __forceinline__ void store(float *state4) const { state4[0] = x1; state4[1] = x2; state4[2] = y1; state4[3] = y2; }
  __device__ __forceinline__ float eval(float x) const { /* products rounded, summed left to right */
    float y = b0 * x;
    y = y + b1 * x1;
    y = y + b2 * x2;
    y = y + a1 * y1;
    y = y + a2 * y2;
    return y;
  }
  __device__ __forceinline__ void commit(float x, float y) { x2 = x1; x1 = x; y2 = y1; y1 = y; }
};
/* one block of one tile row through the cascade, in place; called by all four lanes of the row's quad */
/* LEAN: the form for the kernels that carry blanker / detector code beside it (fewer registers: two workgroups per CU) */
template <bool LEAN = false>
__device__ __forceinline__ void cascade_row(Section &sec, float *row, int s) {
  float yprev = 0.0f, xnext = row[0];
#pragma unroll 4
  for (int i = 0; i < BS + 3; i++) {
    const int n = i - s;
    const float up = dpp_up1(yprev);
    const float xin = xnext;
    xnext = row[i + 1 < BS ? i + 1 : BS - 1]; /* asked for a step ahead: the read's latency passes behind this step's arithmetic
                                               * (the row is also written below, so the compiler will not move the read itself) */
    const float x = s == 0 ? xin : up;
    const float y = sec.eval(x);
    const bool live = n >= 0 && n < BS;
    if (live) {
      sec.commit(x, y);
      yprev = y;
    }
    if constexpr (LEAN) {
      if (live && s == 3) row[n] = y;
    } else {
      row[(live && s == 3) ? n : BS] = y; /* the last section's lane writes the sample; every other lane the row's spare word */
    }
  }
}
/* the oscillator in two passes: the phase recursion alone (one lane per channel: a float add and the wrap), then cosine,
 * sine and the complex product for every sample of the tile in parallel -- they are pure functions of the phase */
__device__ __forceinline__ void phase_row(float &ph, float inc, float *out) {
  for (int t = 0; t < BS; t++) { /* both wrapped candidates are formed and one value selected: as branches the three cases cost
                                  * the lone wave more instructions (exec-mask bookkeeping) than the arithmetic */
    out[t] = ph;
    ph = ph + inc;
    const float down = ph - TWO_PI_F, up = ph + TWO_PI_F;
    ph = ph > TWO_PI_F ? down : (ph < 0.0f ? up : ph);
  }
}
__device__ __forceinline__ void rotate_sample(const float *sine, float ph, float &x, float &y) {
  float pc = (float)((double)ph + 1.5707963267948966);
  if (pc >= TWO_PI_F) pc -= TWO_PI_F;
  if (pc < 0.0f) pc += TWO_PI_F;
  const float c = table_sin(sine, pc);
  float ps = ph >= TWO_PI_F ? ph - TWO_PI_F : ph;
  if (ps < 0.0f) ps += TWO_PI_F;
  const float s = table_sin(sine, ps);
  const float xi = x, yq = y;
  x = fmaf(xi, c, -(s * yq));
  y = fmaf(yq, c, xi * s);
}

/* v / 32767.0 correctly rounded without the division: q0 = v y, r = v - 32767 q0 (exact, fused), q = q0 + r y with
 * y = RN(1 / 32767) -- equal to the IEEE quotient for every int16 v (tests/test_host_logic.py tries all 65 536) */
__device__ __forceinline__ double over_32767(int v) {
  const double y = 1.0 / 32767.0, x = (double)v;
  const double q0 = x * y;
  return fma(fma(-q0, 32767.0, x), y, q0);
}

/* ---- front: conversion, blanker, IF filter, frequency shift (SSB / CW) or the AM / SAM detectors ----------------------
 * 8 channels = 16 tile rows (channel, rail) per workgroup of four waves -- a launch's duration is one workgroup's chain
 * of blocks whatever the grid, so the fewer channels a workgroup carries the shorter it is, down to what the recursions
 * need.  Per block: conversion and the mixer's table work spread over all 256 lanes (element e = lane + 256 j: consecutive
 * lanes on consecutive samples of a row); the cascades with a quad per row on waves 0 and 1 while wave 2 runs the
 * oscillator's phase, one lane per channel; the PLL and the blanker -- true recursions -- on one lane per channel or row. */
constexpr int FW = 256, FCH = 8, PW = 256; /* PW: threads of the pipelined kernels */ /* PW: the pipelined kernels' six waves */
template <bool NB>
__global__ __launch_bounds__(FW, 2) void rdsp_engine_front_kernel(const EngParams p) {
  __shared__ float tf[2 * FCH][PITCH];
  __shared__ float phs[FCH][PITCH];
  __shared__ int locked_of[FCH];
  __shared__ float sine[257];                                     /* the oscillator's table: the PLL reads it twice per sample of a dependent chain */
  const int tid = threadIdx.x, c0 = blockIdx.x * FCH;
  for (int i = tid; i < 257; i += FW) sine[i] = p.sine[i];
  const bool casc = tid < 8 * FCH;                              /* cascade role: section sct of tile row `row` */
  const int row = (tid >> 2) & (2 * FCH - 1), sct = tid & 3;
  const int rch = min(c0 + (row >> 1), p.n_channels - 1);
  const bool row_valid = casc && c0 + (row >> 1) < p.n_channels;
  const bool ssb = p.mode <= 3 || p.mode == 6, am = p.mode == 4 || p.mode == 5;
  Section pre, amf;
  pre.load(p.sets + 20 * p.pre_set + 5 * sct, p.st + (size_t)rch * NF + ST_PRE + 16 * (row & 1) + 4 * sct, (p.resets & RESET_PRE) != 0);
  amf.load(p.sets + 20 * 13 + 5 * sct, p.st + (size_t)rch * NF + ST_AM + 16 * (row & 1) + 4 * sct, false);
  /* serial role: lane 128 + c owns channel c0 + c's scalars (wave 2, beside the cascades' waves 0 and 1) */
  const bool ser = tid >= 128 && tid < 128 + FCH;
  const int sc = (tid - 128) & (FCH - 1);
  const int sch = min(c0 + sc, p.n_channels - 1);
  const bool ser_valid = ser && c0 + sc < p.n_channels;
  float *sst = p.st + (size_t)sch * NF;
  float nco = sst[ST_NCO], amph = sst[ST_AMPH];
  float sam_c = sst[ST_SAM_COS], sam_s = sst[ST_SAM_SIN], sam_u = sst[ST_SAM_U], sam_err = sst[ST_SAM_ERR], sam_hz = sst[ST_SAM_HZ],
        sam_ph = sst[ST_SAM_PH];
  int sam_locked = __float_as_int(sst[ST_SAM_LOCK]);
  /* the blanker's lines: three blocks of I, Q and mask per channel in LDS, slot (n / 128 + nb_base) % 3 holding samples
   * n .. n + 127 of the 384-sample line, so that a block boundary moves a base instead of 768 words */
  constexpr int NBP = NB ? 388 : 1;
  __shared__ float nbl[NB ? 3 : 1][NB ? FCH : 1][NBP];
  __shared__ float nbmag[NB ? FCH : 1][NB ? 180 : 1];
  int nb_base = 0;
  auto nbx = [&](int n) { int sl = (n >> 7) + nb_base; sl = sl >= 3 ? sl - 3 : sl; return sl * 128 + (n & 127); };
  float nb_avg = sst[ST_NB_AVG], nb_last = sst[ST_NB_LAST];
  int nb_hit = __float_as_int(sst[ST_NB_HIT]);
  if constexpr (NB) {
    for (int e = tid; e < 3 * FCH * 384; e += FW) { /* HBM keeps them in line order: I, Q, mask */
      const int k = e / (FCH * 384), r = e - k * (FCH * 384), cl = r / 384, i = r - cl * 384;
      nbl[k][cl][i] = c0 + cl < p.n_channels ? p.nb[(size_t)(c0 + cl) * NB_WORDS + 384 * k + i] : (k == 2 ? 1.0f : 0.0f);
    }
    __syncthreads();
  }
  const float nco_inc = -(p.tuning_offset * RAD_PER_HZ), am_inc = -p.if_centre * RAD_PER_HZ;
  constexpr int EP = FCH * BS / FW; /* elements (int16 pairs, complex samples) per lane and pass */

  for (int b = 0; b < p.n_blocks; b++) {
    for (int j = 0; j < EP; j++) { /* 0xe7b4: / 32767 and the rail's gain, in double; one int16 pair per element */
      const int e = tid + FW * j, cl = e >> 7, t = e & 127;
      const int w = c0 + cl < p.n_channels ? p.iq[(size_t)(c0 + cl) * p.in_stride + (size_t)b * BS + t] : 0;
      tf[2 * cl][t] = (float)(over_32767((int)(int16_t)(w & 0xffff)) * (double)p.gain_i);
      tf[2 * cl + 1][t] = (float)(over_32767(w >> 16) * (double)p.gain_q);
    }
    __syncthreads();
    if constexpr (NB) { /* 0xe14c: two blocks of delay; |I + jQ| against its running average; blanking mask with a taper */
      nb_base = nb_base == 2 ? 0 : nb_base + 1;                    /* the oldest block's slot takes the new one */
      for (int j = 0; j < EP; j++) {
        const int e = tid + FW * j, cl = e >> 7, t = e & 127, at = nbx(256 + t);
        nbl[0][cl][at] = tf[2 * cl][t]; nbl[1][cl][at] = tf[2 * cl + 1][t]; nbl[2][cl][at] = 1.0f;
      }
      __syncthreads();
      for (int e = tid; e < FCH * 178; e += FW) {                  /* the magnitudes are pure functions of the samples */
        const int cl = e / 178, n = 78 + e - cl * 178, at = nbx(n);
        const float vi = nbl[0][cl][at], vq = nbl[1][cl][at];
        nbmag[cl][n - 78] = quick_sqrt1(fmaf(vi, vi, vq * vq));
      }
      __syncthreads();
      if (ser) { /* what is a recursion: the running average and what it decides, then the taper in front of every 0 -> 1 step */
        float *mask = nbl[2][sc];
        nb_hit = 0;
        int zeroed_to = 67;                      /* hits come in rising order: what an earlier one of this pass zeroed stays zero */
        for (int n = 78; n < 256; n++) {
          const float limit = nb_avg * p.nb_ratio;
          nb_last = nbmag[sc][n - 78];
          if (limit < nb_last) {
            if (-p.nb_before <= p.nb_after) {
              const int lo = max(n - p.nb_before, zeroed_to + 1), hi = n + p.nb_after;
              for (int j = lo; j <= hi; j++) mask[nbx(j)] = 0.0f;
              zeroed_to = max(zeroed_to, hi);
            }
            nb_hit = 1;
          }
          nb_avg = fmaf(nb_avg, p.nb_keep, nb_last * p.nb_new);
        }
        const float taper[7] = {0.933f, 0.75f, 0.5f, 0.25f, 0.067f, 0.0f, 0.0f};
        for (int i = 128; i < 256; i++)
          if (mask[nbx(i)] == 1.0f && mask[nbx(i - 1)] == 0.0f)
            for (int j = 0; j < 7; j++) mask[nbx(i - 7 + j)] = taper[j];
      }
      __syncthreads();
      for (int j = 0; j < EP; j++) {
        const int e = tid + FW * j, cl = e >> 7, t = e & 127, at = nbx(t);
        const float mk = nbl[2][cl][at];
        tf[2 * cl][t] = mk * nbl[0][cl][at]; tf[2 * cl + 1][t] = mk * nbl[1][cl][at];
      }
      __syncthreads();
    }
    if (casc) cascade_row<true>(pre, tf[row], sct);
    else if (ser && ssb) phase_row(nco, nco_inc, phs[sc]);       /* 0xe94e: the phase falls by the tuning offset */
    __syncthreads();
    if (ssb) {
      for (int j = 0; j < EP; j++) {
        const int e = tid + FW * j, cl = e >> 7, t = e & 127;
        float x = tf[2 * cl][t], y = tf[2 * cl + 1][t];
        rotate_sample(sine, phs[cl][t], x, y);
        tf[2 * cl][t] = x; tf[2 * cl + 1][t] = y;
      }
      __syncthreads();
      const uint32_t at = p.pos + (uint32_t)b * BS, m = p.ring_size - 1;
      for (int j = 0; j < 2 * EP; j++) { /* into the rings: tile row r is (channel c0 + r / 2, rail r & 1) */
        const int e = tid + FW * j, r = e >> 7, t = e & 127;
        if (c0 + (r >> 1) < p.n_channels)
          ((r & 1) ? p.ring_q : p.ring_i)[(size_t)(c0 + (r >> 1)) * p.ring_size + ((at + (uint32_t)t) & m)] = tf[r][t];
      }
    } else if (am) {
      if (casc) cascade_row<true>(pre, tf[row], sct);                  /* 0xec1c: the IF filter a second time */
      __syncthreads();
      if (p.mode == 5) { /* 0xe390: PLL on the IF signal, one lane per channel */
        if (ser) {
          const float HALF_PI = 1.5707963705062866f, A1 = 0.97239410877227783f, A3 = -0.19194795191287994f;
          float *ri = tf[2 * sc], *rq = tf[2 * sc + 1];
          for (int t = 0; t < BS; t++) {
            const float x = ri[t], q = rq[t];
            const float re = fmaf(x, sam_c, q * sam_s), im = fmaf(q, sam_c, -(sam_s * x));
            float err;
            if (re == 0.0f) err = im > 0.0f ? HALF_PI : (im < 0.0f ? -HALF_PI : 0.0f);
            else if (fabsf(re) > fabsf(im)) {
              const float z = im / re;
              err = fmaf(z, z * A3, A1) * z;
              if (!(re > 0.0f)) err = (float)(im >= 0.0f ? (double)err + 3.1415926535897931 : (double)err - 3.1415926535897931);
            } else {
              const float z = re / im;
              err = fmaf(-z, fmaf(z, z * A3, A1), im > 0.0f ? HALF_PI : -HALF_PI);
            }
            const float u = fmaf(err, p.sam_ga, p.sam_gb * sam_err);
            const double phd = fma((double)(u + sam_u), 0.5, (double)sam_ph);
            sam_hz = fmaf(p.sam_keep, sam_hz, (u * p.sam_hz_per_rad) * p.sam_new);
            sam_ph = (float)phd;
            if ((double)sam_ph >= 3.1415926535897931) sam_ph -= TWO_PI_F;
            if ((double)sam_ph < -3.1415926535897931) sam_ph += TWO_PI_F;
            sam_locked = sam_hz > p.sam_lock_lo ? (sam_hz < p.sam_lock_hi) : 0;
            float pc = (float)((double)sam_ph + 1.5707963267948966);
            if (pc >= TWO_PI_F) pc -= TWO_PI_F;
            if (pc < 0.0f) pc += TWO_PI_F;
            sam_c = table_sin(sine, pc);
            float ps = sam_ph >= TWO_PI_F ? sam_ph - TWO_PI_F : sam_ph;
            if (ps < 0.0f) ps += TWO_PI_F;
            sam_s = table_sin(sine, ps);
            if (sam_locked) {
              ri[t] = fmaf(x, sam_c, q * sam_s);
              rq[t] = fmaf(-x, sam_s, q * sam_c);
            }
            sam_u = u; sam_err = err;
          }
          locked_of[sc] = sam_locked;
        }
      } else if (ser) locked_of[sc] = 0;
      __syncthreads();
      /* AM, and SAM out of lock (0xed02): shift by the IF centre, low-pass, envelope.  A channel in lock keeps the rotated
       * I rail as its audio and none of the detector's state moves */
      if (ser && !locked_of[sc]) phase_row(amph, am_inc, phs[sc]);
      __syncthreads();
      for (int j = 0; j < EP; j++) {
        const int e = tid + FW * j, cl = e >> 7, t = e & 127;
        if (locked_of[cl]) continue;
        float x = tf[2 * cl][t], y = tf[2 * cl + 1][t];
        rotate_sample(sine, phs[cl][t], x, y);
        tf[2 * cl][t] = x; tf[2 * cl + 1][t] = y;
      }
      __syncthreads();
      if (casc) {
        const bool detect = !locked_of[row >> 1];
        const Section keep = amf;
        cascade_row<true>(amf, detect ? tf[row] : phs[row >> 1], sct);   /* the quads of a locked channel run on a row nobody reads ... */
        if (!detect) amf = keep;                                 /* ... and keep their state */
      }
      __syncthreads();
      for (int j = 0; j < EP; j++) {
        const int e = tid + FW * j, cl = e >> 7, t = e & 127;
        if (locked_of[cl]) continue;
        const float x = tf[2 * cl][t], y = tf[2 * cl + 1][t];
        tf[2 * cl][t] = quick_sqrt2(fmaf(x, x, y * y));
      }
      __syncthreads();
      for (int j = 0; j < EP; j++) { /* the demodulated audio is in the I rows */
        const int e = tid + FW * j, cl = e >> 7, t = e & 127;
        if (c0 + cl < p.n_channels) p.audio[(size_t)(c0 + cl) * p.audio_stride + (size_t)b * BS + t] = tf[2 * cl][t];
      }
    }
    __syncthreads();
  }
  if (row_valid) {
    pre.store(p.st + (size_t)rch * NF + ST_PRE + 16 * (row & 1) + 4 * sct);
    if (am) amf.store(p.st + (size_t)rch * NF + ST_AM + 16 * (row & 1) + 4 * sct);
  }
  if (ser_valid) {
    sst[ST_NCO] = nco; sst[ST_AMPH] = amph;
    sst[ST_SAM_COS] = sam_c; sst[ST_SAM_SIN] = sam_s; sst[ST_SAM_U] = sam_u; sst[ST_SAM_ERR] = sam_err; sst[ST_SAM_HZ] = sam_hz;
    sst[ST_SAM_PH] = sam_ph; sst[ST_SAM_LOCK] = __int_as_float(sam_locked);
  }
  if constexpr (NB) {
    if (ser_valid) { sst[ST_NB_AVG] = nb_avg; sst[ST_NB_LAST] = nb_last; sst[ST_NB_HIT] = __int_as_float(nb_hit); }
    __syncthreads();
    for (int e = tid; e < 3 * FCH * 384; e += FW) {
      const int k = e / (FCH * 384), r = e - k * (FCH * 384), cl = r / 384, i = r - cl * 384;
      if (c0 + cl < p.n_channels) p.nb[(size_t)(c0 + cl) * NB_WORDS + 384 * k + i] = nbl[k][cl][nbx(i)];
    }
  }
}

/* ---- the same front stage for the SSB / CW modes without the blanker, as a pipeline of waves ------------------------
 * A lone wave issues an instruction every five cycles or so whatever it depends on, so a block costs its workgroup the SUM
 * of its passes' instruction counts -- unless the passes run on different waves at the same time.  Here they do, each on
 * the block behind the previous one's: waves 2 and 3 convert block s into tile slot s & 3 and rotate / store block s - 2
 * out of slot (s - 2) & 3 (the longest pass of the step: knocking the rotation out cuts 2.2 us of 8.4 per block, knocking
 * the cascade out nothing; six waves per workgroup instead of four ran 1.5 x slower), wave 0 runs the cascades of block s - 1, wave 1 the oscillator's phase of block s - 1; one
 * barrier per step.  A step then lasts as long as its longest pass (the cascade: 131 dependent steps), and the arithmetic
 * of every sample is what it was. */
__global__ __launch_bounds__(PW, 2) void rdsp_engine_front_pipe_kernel(const EngParams p) {
  __shared__ float tf[4][2 * FCH][PITCH];
  __shared__ float phs[4][FCH][PITCH];
  __shared__ float sine[257];                                     /* the oscillator's table beside the data it turns */
  const int tid = threadIdx.x, wave = tid >> 6, c0 = blockIdx.x * FCH;
  for (int i = tid; i < 257; i += PW) sine[i] = p.sine[i];
  __syncthreads();
  const int row = (tid >> 2) & (2 * FCH - 1), sct = tid & 3;      /* wave 0: section sct of tile row `row` */
  const int rch = min(c0 + (row >> 1), p.n_channels - 1);
  Section pre;
  pre.load(p.sets + 20 * p.pre_set + 5 * sct, p.st + (size_t)rch * NF + ST_PRE + 16 * (row & 1) + 4 * sct, (p.resets & RESET_PRE) != 0);
  const int sc = tid & (FCH - 1);                                 /* wave 1, lanes 64 ... 64 + FCH - 1: channel sc's oscillator */
  const bool ser = wave == 1 && (tid & 63) < FCH;
  const int sch = min(c0 + sc, p.n_channels - 1);
  float nco = p.st[(size_t)sch * NF + ST_NCO];
  const float nco_inc = -(p.tuning_offset * RAD_PER_HZ);
  const int wl = tid - 128;                                       /* waves 2 and 3: 128 lanes for the element passes */
  constexpr int EP = FCH * BS / (PW - 128);
  const uint32_t m = p.ring_size - 1;
  for (int step = 0; step < p.n_blocks + 2; step++) {
    if (wave >= 2) {
      if (step < p.n_blocks) { /* 0xe7b4: block `step` comes in */
        float (*t0)[PITCH] = tf[step & 3];
        for (int j = 0; j < EP; j++) {
          const int e = wl + (PW - 128) * j, cl = e >> 7, t = e & 127;
          const int w = c0 + cl < p.n_channels ? p.iq[(size_t)(c0 + cl) * p.in_stride + (size_t)step * BS + t] : 0;
          t0[2 * cl][t] = (float)(over_32767((int)(int16_t)(w & 0xffff)) * (double)p.gain_i);
          t0[2 * cl + 1][t] = (float)(over_32767(w >> 16) * (double)p.gain_q);
        }
      }
      const int b = step - 2;
      if (b >= 0) { /* 0xe94e: block step - 2, filtered and with its phases known, is rotated and leaves for the rings */
        float (*t2)[PITCH] = tf[b & 3];
        const float (*ph)[PITCH] = phs[b & 3];
        const uint32_t at = p.pos + (uint32_t)b * BS;
        for (int j = 0; j < EP; j++) {
          const int e = wl + (PW - 128) * j, cl = e >> 7, t = e & 127;
          float x = t2[2 * cl][t], y = t2[2 * cl + 1][t];
          rotate_sample(sine, ph[cl][t], x, y);
          if (c0 + cl < p.n_channels) {
            const size_t o = (size_t)(c0 + cl) * p.ring_size + ((at + (uint32_t)t) & m);
            p.ring_i[o] = x; p.ring_q[o] = y;
          }
        }
      }
    } else {
      const int b = step - 1;
      if (b >= 0 && b < p.n_blocks) {
        if (wave == 0) cascade_row(pre, tf[b & 3][row], sct);
        else if (ser) phase_row(nco, nco_inc, phs[b & 3][sc]);
      }
    }
    __syncthreads();
  }
  if (wave == 0 && c0 + (row >> 1) < p.n_channels) pre.store(p.st + (size_t)rch * NF + ST_PRE + 16 * (row & 1) + 4 * sct);
  if (ser && c0 + sc < p.n_channels) p.st[(size_t)sch * NF + ST_NCO] = nco;
}

/* ---- 0xea7e: I delayed by 128, Q through the 257-tap Hilbert transformer (odd taps, antisymmetric), side band by sign ---
 * out[t] = sum_k h[k] (q[t - 1 - 2k] - q[t - 255 + 2k]), k = 0 .. 63 in this order, one fused multiply-add each.  An output
 * only meets samples of the other parity, and the outputs t, t + 2, ... meet the same ones shifted by a tap: a lane takes
 * EIGHT outputs of one parity (t = 2 (8 l + j) + p), keeps the two sliding windows in registers, and reads 142 words of
 * LDS for them instead of 1024.  The window of the ring sits in LDS split by parity, index m at m + m / 8: lanes are 8
 * indices apart, so their reads fall 9 words apart (no bank conflicts) and every offset is an immediate.  The delayed I
 * samples come in, and the audio leaves, through a row at pitch 17 for 16 (coalesced 256-byte segments in HBM). */
constexpr int HB_OUT = 2048;                      /* outputs per workgroup */
constexpr int HB_M = (HB_OUT + 256) / 2;          /* samples per parity in the window */
template <int P>
__device__ __forceinline__ void hilbert_eight(const float *par, const float *h, float (&acc)[8]) {
  /* par: the parity array this lane's outputs read (the other parity), already offset by 9 l; P: the outputs' parity */
  float U[8], L[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int ru = j + 127 + P, rl = j + P;
    U[j] = par[ru + (ru >> 3)];
    L[j] = par[rl + (rl >> 3)];
    acc[j] = 0.0f;
  }
#pragma unroll
  for (int k = 0; k < 64; k++) {
    const float hk = h[k];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = fmaf(hk, U[j] - L[j], acc[j]);
    if (k < 63) { /* tap k + 1: the upper window moves one index down, the lower one up */
#pragma unroll
      for (int j = 7; j > 0; j--) U[j] = U[j - 1];
#pragma unroll
      for (int j = 0; j < 7; j++) L[j] = L[j + 1];
      const int ru = 127 + P - (k + 1), rl = 7 + P + (k + 1);
      U[0] = par[ru + (ru >> 3)];
      L[7] = par[rl + (rl >> 3)];
    }
  }
}
__global__ __launch_bounds__(256) void rdsp_engine_hilbert_kernel(const EngParams p) {
  __shared__ float par[2][HB_M + HB_M / 8];
  __shared__ float row[HB_OUT + HB_OUT / 16];
  __shared__ float h[64];
  const int tid = threadIdx.x, ch = blockIdx.y;
  const uint32_t t0 = blockIdx.x * (uint32_t)HB_OUT, m = p.ring_size - 1, n = (uint32_t)p.n_blocks * BS;
  const float *rq = p.ring_q + (size_t)ch * p.ring_size, *ri = p.ring_i + (size_t)ch * p.ring_size;
  for (int i = tid; i < HB_OUT + 256; i += 256) { /* window sample i = t0 - 256 + i, by parity */
    const int mm = i >> 1;
    par[i & 1][mm + (mm >> 3)] = rq[(p.pos + t0 - 256u + (uint32_t)i) & m];
  }
  for (int i = tid; i < HB_OUT; i += 256) row[i + (i >> 4)] = ri[(p.pos + t0 + (uint32_t)i - 128u) & m];
  if (tid < 64) h[tid] = p.hilbert[tid];
  __syncthreads();
  const int P = tid >> 7, lp = tid & 127;             /* waves 0, 1: the even outputs; waves 2, 3: the odd ones */
  float acc[8];
  /* output o = 2 (8 lp + j) + P reads the window at i = o + 255 - 2k and o + 1 + 2k: parity 1 - P, indices
   * 8 lp + j + 127 + P - k and 8 lp + j + P + k */
  if (P == 0) hilbert_eight<0>(par[1] + 9 * lp, h, acc);
  else hilbert_eight<1>(par[0] + 9 * lp, h, acc);
  const bool minus = p.mode == 6 || (p.mode & ~2) == 1;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    float *slot = &row[17 * lp + 2 * j + P];          /* o + o / 16 with o = 16 lp + 2 j + P */
    *slot = minus ? *slot - acc[j] : *slot + acc[j];
  }
  __syncthreads();
  for (int i = tid; i < HB_OUT; i += 256)
    if (t0 + (uint32_t)i < n) p.audio[(size_t)ch * p.audio_stride + t0 + (uint32_t)i] = row[i + (i >> 4)];
}

/* ---- tail: audio band-pass (0xd944), AGC (0xdb58), ALS (0xda24), output (0xebfa) ------------------------------------- */
__device__ __forceinline__ float agc_lookup(const float *curve, float env) {
  const int idx = trunc_s32((double)env * 32767.0);
  int hi = (idx >> 8) & 0xff, hi1;
  if (hi > 127) { hi = 127; hi1 = 128; } else hi1 = hi + 1;
  const float frac = (float)(unsigned)(idx & 0xff) * 0.00390625f;
  const float t0 = curve[hi];
  return fmaf(frac, curve[hi1] - t0, t0);
}

/* 8 channels (16 with the ALS filter) per workgroup of four waves.  Per block: the audio cascade with a quad per channel (wave 0); the AGC's
 * envelope -- the only true recursion in it -- on one lane per channel (wave 1), which leaves for every sample the
 * envelope value its gain is looked up from (or "none yet": the gain carried in); gain, clamp and pack are then pure
 * functions and run on all lanes. */
template <bool ALS>
__global__ __launch_bounds__(FW, 2) void rdsp_engine_tail_kernel(const EngParams p) {
  constexpr int TCH = ALS ? 16 : 8; /* channels per workgroup (measured: 8 is 12 % faster than 16 without the ALS filter, half as fast with it) */
  __shared__ float ta[TCH][PITCH];
  __shared__ float ge[TCH][PITCH];
  __shared__ float curve[130];
  __shared__ float g_in[TCH];
  constexpr int LP = 260;                 /* pitch of a channel's 256-sample ALS line */
  __shared__ float line[ALS ? TCH : 1][ALS ? LP : 1];
  const int tid = threadIdx.x, c0 = blockIdx.x * TCH;
  /* ALS role: wave 1 as 16 quads, quad ac on channel c0 + ac; its lane aq works on every fourth sample */
  const bool als_lane = ALS && (tid >> 6) == 1;
  const int ac = (tid >> 2) & (TCH - 1), aq = tid & 3;
  const int ach = min(c0 + ac, p.n_channels - 1);
  float w[ALS ? ALS_TAPS : 1];
  const bool casc = tid < 4 * TCH;
  const int row = (tid >> 2) & (TCH - 1), sct = tid & 3;
  const int rch = min(c0 + row, p.n_channels - 1);
  Section aud;
  aud.load(p.sets + 20 * p.audio_set + 5 * sct, p.st + (size_t)rch * NF + ST_AUDIO + 4 * sct, (p.resets & RESET_AUDIO) != 0);
  const bool ser = tid >= 64 && tid < 64 + TCH;
  const int sc = (tid - 64) & (TCH - 1);
  const bool ser_valid = ser && c0 + sc < p.n_channels;
  const int sch = min(c0 + sc, p.n_channels - 1);
  float *sst = p.st + (size_t)sch * NF;
  float env = sst[ST_AGC_ENV], g = sst[ST_AGC_GAIN];
  int hang = __float_as_int(sst[ST_AGC_HANG]), active = __float_as_int(sst[ST_AGC_ACTIVE]);
  for (int i = tid; i < 130; i += FW) curve[i] = p.curve[i];
  if constexpr (ALS) {
    if (als_lane) {
      const float *a = p.als + (size_t)ach * ALS_WORDS;
      const bool clear = (p.resets & RESET_ALS) != 0;
      for (int i = aq; i < 256; i += 4) line[ac][i] = clear ? 0.0f : a[i];
#pragma unroll
      for (int k = 0; k < ALS_TAPS; k++) w[k] = clear ? 0.0f : a[256 + k];
    }
  }
  constexpr int EP = TCH * BS / FW;
  for (int b = 0; b < p.n_blocks; b++) {
    __syncthreads();
    for (int j = 0; j < EP; j++) {
      const int e = tid + FW * j, r = e >> 7, t = e & 127;
      ta[r][t] = c0 + r < p.n_channels ? p.audio[(size_t)(c0 + r) * p.audio_stride + (size_t)b * BS + t] : 0.0f;
    }
    __syncthreads();
    if (p.audio_on) {
      if (casc) cascade_row<true>(aud, ta[row], sct);
      __syncthreads();
    }
    if (p.agc_on) {
      if (ser) {
        g_in[sc] = g;
        float last = -1.0f;                      /* the envelope the current gain was looked up from; < 0: none in this block yet */
        const float *a = ta[sc];
        float anext = a[0];
        for (int t = 0; t < BS; t++) {
          float in = fabsf(anext);
          anext = a[t + 1 < BS ? t + 1 : BS - 1];
          if (in > 1.0f) in = 1.0f;
          /* attack (the hang counter is re-armed) / decay (counter at 0) / hold (count down): both candidate envelopes
           * are formed and one is selected -- the channels of a wave are in different states, and as branches every
           * lane would walk all three arms */
          const bool attack = env < in, decay = !attack && hang == 0;
          const float ea = fmaf(env, p.agc_attack_a, in * p.agc_attack_b), ed = fmaf(env, p.agc_decay_a, in * p.agc_decay_b);
          env = attack ? ea : (decay ? ed : env);
          hang = attack ? p.agc_hang_time : (decay ? 0 : hang - 1);
          last = (attack || decay) ? env : last;
          ge[sc][t] = last;
        }
        if (last >= 0.0f) g = agc_lookup(curve, last);
        active = (double)g < 0.98999999999999999;
      }
      __syncthreads();
      for (int j = 0; j < EP; j++) {
        const int e = tid + FW * j, r = e >> 7, t = e & 127;
        const float le = ge[r][t];
        const float gg = le < 0.0f ? g_in[r] : agc_lookup(curve, le);
        float y = (gg * p.agc_makeup) * ta[r][t];
        if (y > 1.0f) y = 1.0f;
        else if (y < -1.0f) y = -1.0f;
        ta[r][t] = y;
      }
      __syncthreads();
    }
    if constexpr (ALS) {
      /* 0xda24: y[n] = sum_k w_k x[n - 3 - k] as a chain of 55 fused multiply-adds, e = x[n] - y; on every fourth sample of a
       * block (its first one included) the taps move by w_k += (e x[n - 3 - k]) / 2.  The chain of one sample cannot be cut,
       * but the four samples between two tap moves see the same taps: the four lanes of a quad take one each (the taps in
       * registers, the same in all four), then every lane makes the move with the fourth lane's error.  Quads of four
       * samples ending on a move: {125 .. 128} (only 128 is this block's), {129 .. 132}, ..., {253 .. 256} (256 is the next
       * block's first: not computed here, no move). */
      if (als_lane) {
        float *x = line[ac], *rowp = ta[ac];
        for (int i = aq; i < 128; i += 4) { x[i] = x[i + 128]; x[i + 128] = rowp[i]; }
        wg_sync<1>();
        for (int g = -1; g < 32; g++) {
          const int n = 129 + 4 * g + aq;
          float y = 0.0f;
#pragma unroll
          for (int k = 0; k < ALS_TAPS; k++) y = fmaf(w[k], x[n - ALS_DELAY - k], y);
          const float err = x[n < 256 ? n : 255] - y;
          if (n >= 128 && n < 256) rowp[n - 128] = p.als_notch ? err : y;
          if (p.als_adaptive && g < 31) {
            const float e3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, err), 0xFF, 0xF, 0xF, false)); /* quad_perm [3,3,3,3] */
            const float *xm = x + (132 + 4 * g) - ALS_DELAY;
#pragma unroll
            for (int k = 0; k < ALS_TAPS; k++) w[k] = fmaf(e3 * xm[-k], 0.5f, w[k]);
          }
        }
      }
      __syncthreads();
    }
    for (int j = 0; j < EP; j++) { /* 0xebfa: x output gain x 32767 toward zero, the low half-word, on both outputs */
      const int e = tid + FW * j, r = e >> 7, t = e & 127;
      const uint32_t v = p.mute ? 0u : ((uint32_t)trunc_s32((double)(ta[r][t] * p.output_gain) * 32767.0) & 0xffffu);
      if (c0 + r < p.n_channels) p.out[(size_t)(c0 + r) * p.out_stride + (size_t)b * BS + t] = (int32_t)(v | (v << 16));
    }
  }
  if (casc && c0 + row < p.n_channels) aud.store(p.st + (size_t)rch * NF + ST_AUDIO + 4 * sct);
  if (ser_valid) {
    sst[ST_AGC_ENV] = env; sst[ST_AGC_GAIN] = g; sst[ST_AGC_HANG] = __int_as_float(hang); sst[ST_AGC_ACTIVE] = __int_as_float(active);
  }
  if constexpr (ALS) {
    if (als_lane && c0 + ac < p.n_channels) {
      float *a = p.als + (size_t)ach * ALS_WORDS;
      for (int i = aq; i < 256; i += 4) a[i] = line[ac][i];
      if (aq == 0) {
#pragma unroll
        for (int k = 0; k < ALS_TAPS; k++) a[256 + k] = w[k];
      }
    }
  }
}

/* The tail stage without the ALS filter as a pipeline of waves (see rdsp_engine_front_pipe_kernel): waves 2 and 3 bring
 * block s in and send block s - 3 out (gain by the curve, clamp, pack), wave 0 runs the audio cascade of block s - 1,
 * wave 1 the AGC's envelope of block s - 2. */
__global__ __launch_bounds__(PW, 2) void rdsp_engine_tail_pipe_kernel(const EngParams p) {
  constexpr int TCH = 8;
  __shared__ float ta[4][TCH][PITCH];
  __shared__ float ge[4][TCH][PITCH];
  __shared__ float g_in[4][TCH];
  __shared__ float curve[130];
  const int tid = threadIdx.x, wave = tid >> 6, c0 = blockIdx.x * TCH;
  const int row = (tid >> 2) & (TCH - 1), sct = tid & 3;
  const bool casc = wave == 0 && tid < 4 * TCH;
  const int rch = min(c0 + row, p.n_channels - 1);
  Section aud;
  aud.load(p.sets + 20 * p.audio_set + 5 * sct, p.st + (size_t)rch * NF + ST_AUDIO + 4 * sct, (p.resets & RESET_AUDIO) != 0);
  const int sc = tid & (TCH - 1);
  const bool ser = wave == 1 && (tid & 63) < TCH;
  const int sch = min(c0 + sc, p.n_channels - 1);
  float *sst = p.st + (size_t)sch * NF;
  float env = sst[ST_AGC_ENV], g = sst[ST_AGC_GAIN];
  int hang = __float_as_int(sst[ST_AGC_HANG]), active = __float_as_int(sst[ST_AGC_ACTIVE]);
  for (int i = tid; i < 130; i += PW) curve[i] = p.curve[i];
  const int wl = tid - 128;
  constexpr int EP = TCH * BS / (PW - 128);
  __syncthreads();
  for (int step = 0; step < p.n_blocks + 3; step++) {
    if (wave >= 2) {
      if (step < p.n_blocks) {
        float (*t0)[PITCH] = ta[step & 3];
        for (int j = 0; j < EP; j++) {
          const int e = wl + (PW - 128) * j, r = e >> 7, t = e & 127;
          t0[r][t] = c0 + r < p.n_channels ? p.audio[(size_t)(c0 + r) * p.audio_stride + (size_t)step * BS + t] : 0.0f;
        }
      }
      const int b = step - 3;
      if (b >= 0) { /* gain, clamp (0xdc10), then 0xebfa: x output gain x 32767 toward zero, the low half-word, on both outputs */
        const float (*t3)[PITCH] = ta[b & 3];
        const float (*e3)[PITCH] = ge[b & 3];
        for (int j = 0; j < EP; j++) {
          const int e = wl + (PW - 128) * j, r = e >> 7, t = e & 127;
          float y = t3[r][t];
          if (p.agc_on) {
            const float le = e3[r][t];
            const float gg = le < 0.0f ? g_in[b & 3][r] : agc_lookup(curve, le);
            y = (gg * p.agc_makeup) * y;
            if (y > 1.0f) y = 1.0f;
            else if (y < -1.0f) y = -1.0f;
          }
          const uint32_t v = p.mute ? 0u : ((uint32_t)trunc_s32((double)(y * p.output_gain) * 32767.0) & 0xffffu);
          if (c0 + r < p.n_channels) p.out[(size_t)(c0 + r) * p.out_stride + (size_t)b * BS + t] = (int32_t)(v | (v << 16));
        }
      }
    } else if (wave == 0) {
      const int b = step - 1;
      if (casc && p.audio_on && b >= 0 && b < p.n_blocks) cascade_row(aud, ta[b & 3][row], sct);
    } else {
      const int b = step - 2;
      if (ser && p.agc_on && b >= 0 && b < p.n_blocks) {
        g_in[b & 3][sc] = g;
        float last = -1.0f;                      /* the envelope the current gain was looked up from; < 0: none in this block yet */
        const float *a = ta[b & 3][sc];
        float *lo = ge[b & 3][sc];
        float anext = a[0];
        for (int t = 0; t < BS; t++) {
          float in = fabsf(anext);
          anext = a[t + 1 < BS ? t + 1 : BS - 1];
          if (in > 1.0f) in = 1.0f;
          const bool attack = env < in, decay = !attack && hang == 0;
          const float ea = fmaf(env, p.agc_attack_a, in * p.agc_attack_b), ed = fmaf(env, p.agc_decay_a, in * p.agc_decay_b);
          env = attack ? ea : (decay ? ed : env);
          hang = attack ? p.agc_hang_time : (decay ? 0 : hang - 1);
          last = (attack || decay) ? env : last;
          lo[t] = last;
        }
        if (last >= 0.0f) g = agc_lookup(curve, last);
        active = (double)g < 0.98999999999999999;
      }
    }
    __syncthreads();
  }
  if (casc && c0 + row < p.n_channels) aud.store(p.st + (size_t)rch * NF + ST_AUDIO + 4 * sct);
  if (ser && c0 + sc < p.n_channels) {
    sst[ST_AGC_ENV] = env; sst[ST_AGC_GAIN] = g; sst[ST_AGC_HANG] = __int_as_float(hang); sst[ST_AGC_ACTIVE] = __int_as_float(active);
  }
}

/* ---- host side ---------------------------------------------------------------------------------------------------- */
float bits_f(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
uint32_t f_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* expf of the C library the engine was linked against (newlib's e_expf.c, Sun's algorithm): the gain curve below is built
 * with it, and a different last bit in one of its 129 entries would be a different gain on every sample that uses it */
float engine_expf(float x) {
  const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, inv_ln2 = 1.4426950216e+00f;
  const float P[5] = {1.6666667163e-01f, -2.7777778450e-03f, 6.6137559770e-05f, -1.6533901999e-06f, 4.1381369442e-08f};
  const uint32_t hx = f_bits(x) & 0x7fffffffu;
  const int neg = (int)(f_bits(x) >> 31);
  if (hx > 0x7f800000u) return x + x;
  if (hx == 0x7f800000u) return neg ? 0.0f : x;
  if (x > 8.8721679688e+01f) return INFINITY;
  if (x < -1.0397208405e+02f) return 0.0f;
  float hi = 0.0f, lo = 0.0f;
  int k = 0;
  if (hx > 0x3eb17218u) {
    if (hx < 0x3F851592u) { hi = neg ? x + ln2_hi : x - ln2_hi; lo = neg ? -ln2_lo : ln2_lo; k = neg ? -1 : 1; }
    else { k = (int)(inv_ln2 * x + (neg ? -0.5f : 0.5f)); const float t = (float)k; hi = x - t * ln2_hi; lo = t * ln2_lo; }
    x = hi - lo;
  } else if (hx < 0x31800000u) return 1.0f + x;
  const float t = x * x;
  const float c = x - t * (P[0] + t * (P[1] + t * (P[2] + t * (P[3] + t * P[4]))));
  if (k == 0) return 1.0f - ((x * c) / (c - 2.0f) - x);
  const float y = 1.0f - ((lo - (x * c) / (2.0f - c)) - hi);
  if (k >= -125) return bits_f(f_bits(y) + ((uint32_t)k << 23));
  return bits_f(f_bits(y) + ((uint32_t)(k + 100) << 23)) * 7.8886090522e-31f;
}
}  // namespace

/* what the sketch's calls set: one set per receiver group (one group = the whole object unless rdsp_engine_set_groups cut it) */
struct EngSettings {
  float input_gain, gain_i, gain_q, iq_balance, output_gain, tuning_offset;
  int mode, mute, audio_on, audio_id, audio_set, pre_set, agc_on, als_on, als_notch, als_adaptive, nb_on, resets;
  float agc_attack_a, agc_attack_b, agc_decay_a, agc_decay_b;
  int agc_hang_time;
  uint32_t pos; /* where the group's next sample goes in its channels' rings (they only move in the SSB / CW modes) */
};
struct rdsp_engine {
  int n_channels, device, max_blocks;
  uint32_t ring_size;
  bool tables;
  float *d_st = nullptr, *d_ring_i = nullptr, *d_ring_q = nullptr, *d_audio = nullptr, *d_nb = nullptr, *d_als = nullptr, *d_tab = nullptr;
  float curve[130], sine[257];
  /* constants of the object (docs/engine.md has their places in the image's AudioSDR) */
  float if_centre, ssb_band, cw_band, agc_makeup, agc_knee_db, agc_slope, agc_threshold_db, sam_ga, sam_gb;
  std::vector<EngSettings> grp; /* at least one */
  std::vector<int> first;       /* first channel of each group, ascending; first[0] = 0 */
  int sel = -1;                 /* the group the setters address; -1: all of them */
};

namespace {
constexpr size_t TAB_SETS = 0, TAB_HILBERT = 300, TAB_SINE = 364, TAB_CURVE = 621, TAB_WORDS = 751;

void engine_agc_curve(rdsp_engine_t *e) { /* 0xdd40: soft-knee compressor curve over the envelope, 1/128 per entry */
  const double ln10ish = 2.3025, db_per_octave = 6.026; /* the library's own constants */
  const double T = (double)e->agc_threshold_db, W = (double)e->agc_knee_db;
  const float x_lo = engine_expf((float)(((T - W * 0.5) * ln10ish) / 20.0)), x_hi = engine_expf((float)(((T + W * 0.5) * ln10ish) / 20.0));
  for (int i = 0; i < 130; i++) {
    const float x = (float)i * 0.0078125f;
    if (x_lo > x) { e->curve[i] = 1.0f; continue; }
    int ex;
    const float m = frexpf(x, &ex);
    const float log2x = fmaf(m, fmaf(m, fmaf(m, 1.2314958572387695f, -4.1185250282287598f), 6.021970272064209f), -3.1339645385742188f) + (float)ex;
    const float xdb = (float)((double)log2x * db_per_octave);
    float gdb;
    if (x_hi >= x) {
      const double d = fma(W, 0.5, (double)(xdb - e->agc_threshold_db));
      gdb = (float)(((((double)e->agc_slope - 1.0) * d) * d) / (W + W) + (double)xdb) - xdb;
    } else {
      gdb = fmaf(xdb - e->agc_threshold_db, e->agc_slope, e->agc_threshold_db) - xdb;
    }
    e->curve[i] = engine_expf((float)(((double)gdb * ln10ish) / 20.0));
  }
}
void engine_sam_constants(rdsp_engine_t *e) { /* 0xed34 with the constructor's loop parameters */
  const float wn = bits_f(0x3e50fac7), zeta = 2.0f, kd = 1.0f, ko = 1.0f;
  const double k4 = (double)(1.0f / (kd * ko)) * 4.0, den = 1.0 / ((double)zeta * 4.0) + (double)zeta;
  const float g1 = (float)((k4 * (double)zeta * (double)wn) / den), g2 = (float)((k4 * (double)wn * (double)wn) / (den * den));
  e->sam_ga = g1 + g2;
  e->sam_gb = g2;
}
int engine_fail(const char *what, hipError_t err) {
  rdsp_set_error("%s: %s", what, hipGetErrorString(err));
  return RDSP_ERR_HIP;
}
/* the setters address the selected group, or all of them */
template <typename F>
int for_selected(rdsp_engine_t *e, F f) {
  if (!e) return RDSP_ERR_INVALID;
  for (size_t g = 0; g < e->grp.size(); g++)
    if (e->sel < 0 || (size_t)e->sel == g) f(e->grp[g]);
  return RDSP_OK;
}
void settings_agc_mode(EngSettings &s, int mode) { /* 0xdfe0 */
  static const uint32_t k[4][4] = {{0, 0, 0, 0}, {0x3f79673b, 0x3cd318a0, 0x3f7fddca, 0x3a08d800}, {0x3f7d5732, 0x3c2a3380, 0x3f7ff250, 0x395b0000},
                                   {0x3f7eaab6, 0x3baaa500, 0x3f7ff928, 0x38db0000}};
  static const int hang[4] = {0, 4410, 22050, 88200};
  if (mode == 0) { s.agc_on = 0; return; }
  if (mode < 0 || mode > 3) return; /* the engine ignores other values */
  s.agc_attack_a = bits_f(k[mode][0]); s.agc_attack_b = bits_f(k[mode][1]);
  s.agc_decay_a = bits_f(k[mode][2]); s.agc_decay_b = bits_f(k[mode][3]);
  s.agc_hang_time = hang[mode];
  s.agc_on = 1;
}
void settings_demod(const rdsp_engine_t *e, EngSettings &s, int mode) { /* 0xd798 */
  s.mode = mode & 0xffff;
  switch (s.mode) {
    case 0: s.tuning_offset = (float)((double)e->if_centre + (double)e->ssb_band * 0.5); s.pre_set = 12; break;
    case 1: s.tuning_offset = (float)((double)e->if_centre - (double)e->ssb_band * 0.5); s.pre_set = 12; break;
    case 6: s.tuning_offset = (float)((double)e->if_centre - (double)e->ssb_band * 0.5); s.pre_set = 11; break;
    case 2: s.tuning_offset = (float)((double)e->if_centre + (double)e->cw_band * 0.5); s.pre_set = 10; break;
    case 3: s.tuning_offset = (float)((double)e->if_centre - (double)e->cw_band * 0.5); s.pre_set = 10; break;
    case 4: case 5: s.tuning_offset = e->if_centre; s.pre_set = 14; break;
    default: return;
  }
  s.resets |= RESET_PRE; /* arm_biquad_cascade_df1_init_f32 clears the state */
}
EngSettings settings_as_constructed(const rdsp_engine_t *e) { /* AudioSDR::AudioSDR (0x6744) and its init (0xede4) */
  EngSettings s;
  memset(&s, 0, sizeof s);
  s.input_gain = s.gain_i = s.gain_q = s.iq_balance = s.output_gain = 1.0f;
  s.audio_set = 3; s.nb_on = 1; s.als_notch = 1; s.als_adaptive = 1;
  settings_agc_mode(s, 2); /* 0xdf14: the medium attack with the slow decay and the fast hang time */
  s.agc_decay_a = bits_f(0x3f7ff928); s.agc_decay_b = bits_f(0x38db0000); s.agc_hang_time = 4410;
  settings_demod(e, s, 0);
  s.resets = 0;
  return s;
}
}  // namespace

extern "C" {

int rdsp_engine_setAGCmode(rdsp_engine_t *e, int mode) { return for_selected(e, [&](EngSettings &s) { settings_agc_mode(s, mode); }); }
int rdsp_engine_enableAGC(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.agc_on = 1; }); } /* 0xdfd4 */
float rdsp_engine_setDemodMode(rdsp_engine_t *e, int mode) {
  if (!e) return 0.0f;
  (void)for_selected(e, [&](EngSettings &s) { settings_demod(e, s, mode); });
  return e->grp[e->sel < 0 ? 0 : (size_t)e->sel].tuning_offset;
}
int rdsp_engine_setAudioFilter(rdsp_engine_t *e, int id) { /* 0xd97c */
  static const int set_of_id[10] = {7, 8, 9, 0, 1, 2, 3, 4, 5, 6};
  return for_selected(e, [&](EngSettings &s) {
    if (id == 10) s.audio_on = 0;
    else if (id >= 0 && id < 10) { s.audio_set = set_of_id[id]; s.resets |= RESET_AUDIO; }
    s.audio_id = id;
  });
}
int rdsp_engine_enableAudioFilter(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.audio_on = 1; }); }
int rdsp_engine_setInputGain(rdsp_engine_t *e, float g) { /* 0xd8a0 */
  if (g > 10.0f) g = 10.0f;
  else if (g < 0.0f) g = 0.0f;
  return for_selected(e, [&](EngSettings &s) { s.input_gain = g; s.gain_i = s.iq_balance * g; s.gain_q = g; });
}
int rdsp_engine_setIQgainBalance(rdsp_engine_t *e, float b) { /* 0xd8f0 */
  return for_selected(e, [&](EngSettings &s) { s.iq_balance = b; s.gain_i = b * s.input_gain; s.gain_q = s.input_gain; });
}
int rdsp_engine_setOutputGain(rdsp_engine_t *e, float g) { return for_selected(e, [&](EngSettings &s) { s.output_gain = g; }); }
int rdsp_engine_setMute(rdsp_engine_t *e, int on) { return for_selected(e, [&](EngSettings &s) { s.mute = on ? 1 : 0; }); }
int rdsp_engine_enableALSfilter(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.als_on = 1; s.resets |= RESET_ALS; }); }
int rdsp_engine_disableALSfilter(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.als_on = 0; }); }
int rdsp_engine_setALSfilterNotch(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.als_notch = 1; }); }
int rdsp_engine_setALSfilterPeak(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.als_notch = 0; }); }
int rdsp_engine_setALSfilterAdaptive(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.als_adaptive = 1; }); }
int rdsp_engine_enableNoiseBlanker(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.nb_on = 1; }); }
int rdsp_engine_disableNoiseBlanker(rdsp_engine_t *e) { return for_selected(e, [](EngSettings &s) { s.nb_on = 0; }); }
int rdsp_engine_channels(const rdsp_engine_t *e) { return e ? e->n_channels : 0; }
int rdsp_engine_device(const rdsp_engine_t *e) { return e ? e->device : -1; }
int rdsp_engine_max_blocks(const rdsp_engine_t *e) { return e ? e->max_blocks : 0; }
const float *rdsp_engine_agc_curve(const rdsp_engine_t *e) { return e ? e->curve : nullptr; }
const float *rdsp_engine_sine_table(const rdsp_engine_t *e) { return e ? e->sine : nullptr; }

/* Receiver groups: the sketch has ONE receiver, so one mode, one audio filter, one AGC setting; an object of many channels
 * can be cut into groups of consecutive channels that each carry their own.  first_channel[g] is group g's first channel
 * (ascending, first_channel[0] = 0); new groups start as copies of the group their first channel was in.  The setters
 * address the group chosen with rdsp_engine_select_group (-1, the default: every group).  A call of rdsp_engine_update
 * launches each group's kernels on its channel range; the signal state of a channel does not care which group it is in. */
int rdsp_engine_set_groups(rdsp_engine_t *e, int n_groups, const int *first_channel) {
  if (!e || n_groups < 1 || !first_channel || first_channel[0] != 0) return RDSP_ERR_INVALID;
  for (int g = 1; g < n_groups; g++)
    if (first_channel[g] <= first_channel[g - 1] || first_channel[g] >= e->n_channels) return RDSP_ERR_INVALID;
  std::vector<EngSettings> grp((size_t)n_groups);
  for (int g = 0; g < n_groups; g++) {
    size_t from = 0;
    while (from + 1 < e->first.size() && e->first[from + 1] <= first_channel[g]) from++;
    grp[(size_t)g] = e->grp[from];
  }
  e->grp.swap(grp);
  e->first.assign(first_channel, first_channel + n_groups);
  e->sel = -1;
  return RDSP_OK;
}
int rdsp_engine_groups(const rdsp_engine_t *e) { return e ? (int)e->grp.size() : 0; }
int rdsp_engine_select_group(rdsp_engine_t *e, int group) {
  if (!e || group < -1 || group >= (int)e->grp.size()) return RDSP_ERR_INVALID;
  e->sel = group;
  return RDSP_OK;
}

void rdsp_engine_destroy(rdsp_engine_t *e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (float *p : {e->d_st, e->d_ring_i, e->d_ring_q, e->d_audio, e->d_nb, e->d_als, e->d_tab})
    if (p) (void)hipFree(p);
  delete e;
}

/* device state as AudioSDR::AudioSDR (0x6744) + its init (0xede4) leave it: lines and filter states zero, the blanker's
 * mask lines 1.0, its running average 10.0, the PLL's frequency estimate 1890 Hz */
int rdsp_engine_reset(rdsp_engine_t *e, void *stream) {
  if (!e) return RDSP_ERR_INVALID;
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipSetDevice(e->device);
  const size_t n = (size_t)e->n_channels;
  std::vector<float> st(n * NF, 0.0f), nb(n * NB_WORDS, 0.0f);
  for (size_t c = 0; c < n; c++) {
    st[c * NF + ST_SAM_HZ] = 1890.0f;
    st[c * NF + ST_NB_AVG] = 10.0f;
    st[c * NF + ST_AGC_ACTIVE] = bits_f(1u); /* the flag's value until the AGC first runs */
    for (int i = 0; i < 384; i++) nb[c * NB_WORDS + 768 + i] = 1.0f;
  }
  if (err == hipSuccess) err = hipMemcpyAsync(e->d_st, st.data(), st.size() * 4, hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipMemcpyAsync(e->d_nb, nb.data(), nb.size() * 4, hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipMemsetAsync(e->d_ring_i, 0, n * e->ring_size * 4, s);
  if (err == hipSuccess) err = hipMemsetAsync(e->d_ring_q, 0, n * e->ring_size * 4, s);
  if (err == hipSuccess) err = hipMemsetAsync(e->d_als, 0, n * ALS_WORDS * 4, s);
  if (err == hipSuccess) err = hipStreamSynchronize(s); /* the host vectors go away */
  for (auto &g : e->grp) { g.pos = 0; g.resets = 0; }
  return err == hipSuccess ? RDSP_OK : engine_fail("rdsp_engine_reset", err);
}

int rdsp_engine_create(int n_channels, int device, int max_blocks_per_call, rdsp_engine_t **out) {
  if (!out || n_channels < 1 || max_blocks_per_call < 1 || max_blocks_per_call > 4096) {
    rdsp_set_error("rdsp_engine_create: bad argument");
    return RDSP_ERR_INVALID;
  }
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
    rdsp_set_error("rdsp_engine_create: no HIP device (this library has no CPU path)");
    return RDSP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count || hipSetDevice(device) != hipSuccess) {
    rdsp_set_error("rdsp_engine_create: device %d of %d", device, count);
    return RDSP_ERR_INVALID;
  }
  rdsp_engine_t *e = new rdsp_engine();
  e->n_channels = n_channels; e->device = device; e->max_blocks = max_blocks_per_call;
  e->ring_size = 512;
  while (e->ring_size < (uint32_t)max_blocks_per_call * BS + 256u) e->ring_size <<= 1;
  e->tables = false;
  /* the constructor's values */
  e->if_centre = 6890.0f; e->ssb_band = 3000.0f; e->cw_band = 1000.0f;
  e->agc_makeup = 10.0f; e->agc_threshold_db = -60.0f; e->agc_slope = bits_f(0x3dcccccd); e->agc_knee_db = 2.0f;
  engine_agc_curve(e);
  engine_sam_constants(e);
  for (int k = 0; k < 257; k++) e->sine[k] = (float)(round(sin(2.0 * 3.14159265358979323846 * k / 256.0) * 1e8) / 1e8);
  e->grp.assign(1, settings_as_constructed(e));
  e->first.assign(1, 0);
  const size_t n = (size_t)n_channels;
  hipError_t err = hipMalloc((void **)&e->d_st, n * NF * 4);
  if (err == hipSuccess) err = hipMalloc((void **)&e->d_ring_i, n * e->ring_size * 4);
  if (err == hipSuccess) err = hipMalloc((void **)&e->d_ring_q, n * e->ring_size * 4);
  if (err == hipSuccess) err = hipMalloc((void **)&e->d_audio, n * (size_t)max_blocks_per_call * BS * 4);
  if (err == hipSuccess) err = hipMalloc((void **)&e->d_nb, n * NB_WORDS * 4);
  if (err == hipSuccess) err = hipMalloc((void **)&e->d_als, n * ALS_WORDS * 4);
  if (err == hipSuccess) err = hipMalloc((void **)&e->d_tab, TAB_WORDS * 4);
  if (err != hipSuccess) {
    rdsp_engine_destroy(e);
    rdsp_set_error("rdsp_engine_create: %s", hipGetErrorString(err));
    return RDSP_ERR_NOMEM;
  }
  const int rc = rdsp_engine_reset(e, nullptr);
  if (rc != RDSP_OK) { rdsp_engine_destroy(e); return rc; }
  *out = e;
  return RDSP_OK;
}

/* the engine's coefficient tables: fifteen sets of four {b0, b1, b2, a1, a2} sections in the image's order (ten audio
 * band-passes, then the IF filters: CW, mode 6, SSB, the AM detector's low-pass, AM) and the 64 taps of one side of the
 * Hilbert transformer, outermost first */
int rdsp_engine_load_tables(rdsp_engine_t *e, const float *biquad_sets15x20, const float *hilbert64) {
  if (!e || !biquad_sets15x20 || !hilbert64) return RDSP_ERR_INVALID;
  std::vector<float> t(TAB_WORDS);
  memcpy(&t[TAB_SETS], biquad_sets15x20, 300 * 4);
  memcpy(&t[TAB_HILBERT], hilbert64, 64 * 4);
  memcpy(&t[TAB_SINE], e->sine, 257 * 4);
  memcpy(&t[TAB_CURVE], e->curve, 130 * 4);
  hipError_t err = hipSetDevice(e->device);
  if (err == hipSuccess) err = hipMemcpy(e->d_tab, t.data(), TAB_WORDS * 4, hipMemcpyHostToDevice);
  if (err != hipSuccess) return engine_fail("rdsp_engine_load_tables", err);
  e->tables = true;
  return RDSP_OK;
}

/* AudioSDR::update (0xe730) for n_blocks consecutive 128-sample blocks of every channel.  d_iq: [ch][t] int16 pairs
 * (I, Q), in_stride pairs from one channel's row to the next; d_lr: [ch][t] int16 pairs, the engine's two outputs
 * (it transmits the same block on both, INO:81-86) */
int rdsp_engine_update(rdsp_engine_t *e, const int16_t *d_iq, size_t in_stride, int n_blocks, int16_t *d_lr, size_t out_stride, void *stream) {
  if (!e || !d_iq || !d_lr || n_blocks < 0 || n_blocks > e->max_blocks || in_stride < (size_t)n_blocks * BS || out_stride < (size_t)n_blocks * BS) {
    rdsp_set_error("rdsp_engine_update: bad argument (n_blocks %d of at most %d)", n_blocks, e ? e->max_blocks : 0);
    return RDSP_ERR_INVALID;
  }
  if (!e->tables) {
    rdsp_set_error("rdsp_engine_update: the engine's coefficient tables are not loaded (rdsp_engine_load_tables)");
    return RDSP_ERR_NOT_READY;
  }
  if (n_blocks == 0) return RDSP_OK;
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipSetDevice(e->device);
  if (err != hipSuccess) return engine_fail("rdsp_engine_update", err);
  const size_t audio_stride = (size_t)e->max_blocks * BS;
  for (size_t g = 0; g < e->grp.size(); g++) {
    EngSettings &q = e->grp[g];
    const int c0 = e->first[g], n = (g + 1 < e->grp.size() ? e->first[g + 1] : e->n_channels) - c0;
    EngParams p;
    memset(&p, 0, sizeof p);
    p.iq = (const int32_t *)d_iq + (size_t)c0 * in_stride; p.in_stride = in_stride;
    p.out = (int32_t *)d_lr + (size_t)c0 * out_stride; p.out_stride = out_stride;
    p.n_channels = n; p.n_blocks = n_blocks; p.st = e->d_st + (size_t)c0 * NF;
    p.ring_i = e->d_ring_i + (size_t)c0 * e->ring_size; p.ring_q = e->d_ring_q + (size_t)c0 * e->ring_size;
    p.ring_size = e->ring_size; p.pos = q.pos;
    p.audio = e->d_audio + (size_t)c0 * audio_stride; p.audio_stride = audio_stride;
    p.nb = e->d_nb + (size_t)c0 * NB_WORDS; p.als = e->d_als + (size_t)c0 * ALS_WORDS;
    p.sets = e->d_tab + TAB_SETS; p.hilbert = e->d_tab + TAB_HILBERT; p.sine = e->d_tab + TAB_SINE; p.curve = e->d_tab + TAB_CURVE;
    p.mode = q.mode; p.mute = q.mute; p.audio_on = q.audio_on; p.agc_on = q.agc_on; p.als_notch = q.als_notch;
    p.als_adaptive = q.als_adaptive; p.resets = q.resets; p.pre_set = q.pre_set; p.audio_set = q.audio_set;
    p.gain_i = q.gain_i; p.gain_q = q.gain_q; p.output_gain = q.output_gain; p.tuning_offset = q.tuning_offset; p.if_centre = e->if_centre;
    p.agc_attack_a = q.agc_attack_a; p.agc_attack_b = q.agc_attack_b; p.agc_decay_a = q.agc_decay_a; p.agc_decay_b = q.agc_decay_b;
    p.agc_makeup = e->agc_makeup; p.agc_hang_time = q.agc_hang_time;
    p.nb_keep = 0.995f; p.nb_new = bits_f(0x3ba3d700); p.nb_ratio = 1.2f; p.nb_before = 10; p.nb_after = 10;
    p.sam_keep = 0.995f; p.sam_new = bits_f(0x3ba3d700); p.sam_hz_per_rad = bits_f(0x45db55dd); p.sam_lock_lo = 3890.0f; p.sam_lock_hi = 9890.0f;
    p.sam_ga = e->sam_ga; p.sam_gb = e->sam_gb;
    const bool ssb = q.mode <= 3 || q.mode == 6;
    const int tch = q.als_on ? 16 : 8;
    const dim3 gf((unsigned)((n + FCH - 1) / FCH)), gt((unsigned)((n + tch - 1) / tch));
    if (q.nb_on) hipLaunchKernelGGL(rdsp_engine_front_kernel<true>, gf, dim3(FW), 0, s, p);
    else if (ssb) hipLaunchKernelGGL(rdsp_engine_front_pipe_kernel, gf, dim3(PW), 0, s, p);
    else hipLaunchKernelGGL(rdsp_engine_front_kernel<false>, gf, dim3(FW), 0, s, p);
    if (ssb) { /* (a mode number the engine does not know leaves its audio buffer as it was: the last call's) */
      const dim3 gh((unsigned)((n_blocks * BS + HB_OUT - 1) / HB_OUT), (unsigned)n);
      hipLaunchKernelGGL(rdsp_engine_hilbert_kernel, gh, dim3(256), 0, s, p);
    }
    if (q.als_on) hipLaunchKernelGGL(rdsp_engine_tail_kernel<true>, gt, dim3(FW), 0, s, p);
    else hipLaunchKernelGGL(rdsp_engine_tail_pipe_kernel, gt, dim3(PW), 0, s, p);
    err = hipGetLastError();
    if (err != hipSuccess) return engine_fail("rdsp_engine_update launch", err);
    if (ssb) q.pos = (q.pos + (uint32_t)n_blocks * BS) & (e->ring_size - 1); /* the lines only move when the SSB / CW path runs */
    q.resets = 0;
  }
  return RDSP_OK;
}

/* ---- the signal state of a channel range as data: resume, or move receivers between objects / GPUs ---------------------
 * Blob = header {magic, version, n_channels} + per channel: the 96 state words, the last 512 samples of both lines of the
 * side-band network in time order (whatever the ring's size and position here or there), the blanker's lines, the ALS
 * filter's line and taps.  Settings are not part of it (they belong to the group the channels land in). */
namespace {
constexpr uint32_t STATE_MAGIC = 0x45534452u; /* "RDSE" */
constexpr size_t STATE_CH_WORDS = NF + 1024 + NB_WORDS + ALS_WORDS;
uint32_t pos_of_channel(const rdsp_engine_t *e, int ch) {
  size_t g = 0;
  while (g + 1 < e->first.size() && e->first[g + 1] <= ch) g++;
  return e->grp[g].pos;
}
}  // namespace
size_t rdsp_engine_state_bytes(const rdsp_engine_t *e, int n_channels) {
  return (e && n_channels > 0) ? 16 + (size_t)n_channels * STATE_CH_WORDS * 4 : 0;
}
int rdsp_engine_save_state(rdsp_engine_t *e, int first_channel, int n_channels, void *host_buf, size_t bytes, void *stream) {
  if (!e || !host_buf || first_channel < 0 || n_channels < 1 || first_channel + n_channels > e->n_channels ||
      bytes < rdsp_engine_state_bytes(e, n_channels)) {
    rdsp_set_error("rdsp_engine_save_state: bad argument");
    return RDSP_ERR_INVALID;
  }
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipSetDevice(e->device);
  const size_t n = (size_t)n_channels, c0 = (size_t)first_channel, R = e->ring_size;
  std::vector<float> st(n * NF), ri(n * R), rq(n * R), nb(n * NB_WORDS), als(n * ALS_WORDS);
  if (err == hipSuccess) err = hipMemcpyAsync(st.data(), e->d_st + c0 * NF, st.size() * 4, hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipMemcpyAsync(ri.data(), e->d_ring_i + c0 * R, ri.size() * 4, hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipMemcpyAsync(rq.data(), e->d_ring_q + c0 * R, rq.size() * 4, hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipMemcpyAsync(nb.data(), e->d_nb + c0 * NB_WORDS, nb.size() * 4, hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipMemcpyAsync(als.data(), e->d_als + c0 * ALS_WORDS, als.size() * 4, hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  if (err != hipSuccess) return engine_fail("rdsp_engine_save_state", err);
  uint32_t *hdr = (uint32_t *)host_buf;
  hdr[0] = STATE_MAGIC; hdr[1] = 1; hdr[2] = (uint32_t)n_channels; hdr[3] = 0;
  float *w = (float *)(hdr + 4);
  for (size_t c = 0; c < n; c++, w += STATE_CH_WORDS) {
    const uint32_t pos = pos_of_channel(e, first_channel + (int)c);
    memcpy(w, &st[c * NF], NF * 4);
    for (uint32_t i = 0; i < 512; i++) { /* sample pos - 512 + i */
      w[NF + i] = ri[c * R + ((pos - 512u + i) & (uint32_t)(R - 1))];
      w[NF + 512 + i] = rq[c * R + ((pos - 512u + i) & (uint32_t)(R - 1))];
    }
    memcpy(w + NF + 1024, &nb[c * NB_WORDS], NB_WORDS * 4);
    memcpy(w + NF + 1024 + NB_WORDS, &als[c * ALS_WORDS], ALS_WORDS * 4);
  }
  return RDSP_OK;
}
int rdsp_engine_load_state(rdsp_engine_t *e, int first_channel, const void *host_buf, size_t bytes, void *stream) {
  const uint32_t *hdr = (const uint32_t *)host_buf;
  if (!e || !host_buf || bytes < 16 || hdr[0] != STATE_MAGIC || hdr[1] != 1) {
    rdsp_set_error("rdsp_engine_load_state: not an engine state blob of this version");
    return RDSP_ERR_INVALID;
  }
  const size_t n = hdr[2], c0 = (size_t)first_channel, R = e->ring_size;
  if (first_channel < 0 || n < 1 || c0 + n > (size_t)e->n_channels || bytes < 16 + n * STATE_CH_WORDS * 4) {
    rdsp_set_error("rdsp_engine_load_state: %zu channels at %d do not fit", n, first_channel);
    return RDSP_ERR_INVALID;
  }
  std::vector<float> st(n * NF), ri(n * R, 0.0f), rq(n * R, 0.0f), nb(n * NB_WORDS), als(n * ALS_WORDS);
  const float *w = (const float *)(hdr + 4);
  for (size_t c = 0; c < n; c++, w += STATE_CH_WORDS) {
    const uint32_t pos = pos_of_channel(e, first_channel + (int)c);
    memcpy(&st[c * NF], w, NF * 4);
    for (uint32_t i = 0; i < 512; i++) {
      ri[c * R + ((pos - 512u + i) & (uint32_t)(R - 1))] = w[NF + i];
      rq[c * R + ((pos - 512u + i) & (uint32_t)(R - 1))] = w[NF + 512 + i];
    }
    memcpy(&nb[c * NB_WORDS], w + NF + 1024, NB_WORDS * 4);
    memcpy(&als[c * ALS_WORDS], w + NF + 1024 + NB_WORDS, ALS_WORDS * 4);
  }
  hipStream_t s = (hipStream_t)stream;
  hipError_t err = hipSetDevice(e->device);
  if (err == hipSuccess) err = hipMemcpyAsync(e->d_st + c0 * NF, st.data(), st.size() * 4, hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipMemcpyAsync(e->d_ring_i + c0 * R, ri.data(), ri.size() * 4, hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipMemcpyAsync(e->d_ring_q + c0 * R, rq.data(), rq.size() * 4, hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipMemcpyAsync(e->d_nb + c0 * NB_WORDS, nb.data(), nb.size() * 4, hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipMemcpyAsync(e->d_als + c0 * ALS_WORDS, als.data(), als.size() * 4, hipMemcpyHostToDevice, s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  return err == hipSuccess ? RDSP_OK : engine_fail("rdsp_engine_load_state", err);
}

/* per-channel scalars for tests and monitoring: [n_channels][8] = oscillator phase, AGC gain, AGC envelope, hang counter,
 * AGC-active flag, PLL frequency estimate (Hz), PLL lock flag, blanker-hit flag */
int rdsp_engine_get_scalars(rdsp_engine_t *e, float *host_out, void *stream) {
  if (!e || !host_out) return RDSP_ERR_INVALID;
  std::vector<float> st((size_t)e->n_channels * NF);
  hipError_t err = hipSetDevice(e->device);
  if (err == hipSuccess) err = hipMemcpyAsync(st.data(), e->d_st, st.size() * 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
  if (err != hipSuccess) return engine_fail("rdsp_engine_get_scalars", err);
  for (int c = 0; c < e->n_channels; c++) {
    const float *s = &st[(size_t)c * NF];
    float *o = host_out + (size_t)c * 8;
    int hang, active, lock, hit;
    memcpy(&hang, &s[ST_AGC_HANG], 4); memcpy(&active, &s[ST_AGC_ACTIVE], 4); memcpy(&lock, &s[ST_SAM_LOCK], 4); memcpy(&hit, &s[ST_NB_HIT], 4);
    o[0] = s[ST_NCO]; o[1] = s[ST_AGC_GAIN]; o[2] = s[ST_AGC_ENV]; o[3] = (float)hang; o[4] = (float)active; o[5] = s[ST_SAM_HZ];
    o[6] = (float)lock; o[7] = (float)hit;
  }
  return RDSP_OK;
}

}  // extern "C"

/* ==== `AudioSDRpreProcessor preProcessor;` (INO:53, wired INO:71-72, :117-118; image ::update 0xee88) ================
 * The I2S input of the Teensy can start with one rail a sample late.  While detection is on, every block goes through a
 * 128-point complex FFT; if the strongest of bins 5 ... 122 stands more than 10 x above their mean and its mirror image
 * is less than 20 dB down, a bad-count rises; at the eleventh bad block in a row the remedy moves on (none -> I one
 * sample later -> Q one sample later -> none); after 1000 counted blocks detection switches itself off.  swapIQ
 * exchanges the rails on the way out.  One wave per channel: the transform is a radix-2 pass per LDS exchange, two
 * points per lane; the scan over the bins, whose order the sums depend on, is the first lane's.  Blocks of a call are
 * taken in order, since a block's verdict decides how the next one is read. */
namespace {
struct PreParams {
  const int32_t *iq; size_t in_stride; int32_t *out; size_t out_stride;
  int n_channels, n_blocks, swap, restart;
  int16_t *st; /* [ch][6]: slip, saved sample, bad count, counted blocks, detecting, pad */
  const float *tw; /* 64 x (cos, sin) of -2 pi k / 128 */
};
__global__ __launch_bounds__(64) void rdsp_preproc_kernel(const PreParams p) {
  __shared__ float re[128], im[128];
  __shared__ int16_t raw[2][129];
  __shared__ int verdict[4];
  const int lane = threadIdx.x, ch = blockIdx.x;
  int16_t *st = p.st + (size_t)ch * 6;
  int slip = p.restart ? 0 : st[0], saved = st[1], bad = p.restart ? 0 : st[2], checks = p.restart ? 0 : st[3], detect = p.restart ? 1 : st[4];
  const int32_t *src = p.iq + (size_t)ch * p.in_stride;
  int32_t *dst = p.out + (size_t)ch * p.out_stride;
  for (int b = 0; b < p.n_blocks; b++) {
    const int w0 = src[(size_t)b * BS + lane], w1 = src[(size_t)b * BS + lane + 64];
    raw[0][lane + 1] = (int16_t)(w0 & 0xffff); raw[1][lane + 1] = (int16_t)(w0 >> 16);
    raw[0][lane + 65] = (int16_t)(w1 & 0xffff); raw[1][lane + 65] = (int16_t)(w1 >> 16);
    wg_sync<1>();
    if (slip != 0) { /* the late rail is read one place to the left; its last sample waits for the next block.  As compiled
                      * (0xefe8 stores through the I block's pointer in both cases), the carried sample always lands in
                      * I[0]: with Q delayed, Q[0] keeps the block's own first sample */
      const int r = slip == 1 ? 0 : 1;
      const int last = raw[r][128];
      if (lane == 0) { raw[0][slip == 1 ? 0 : 1] = (int16_t)saved; if (slip == -1) raw[1][0] = raw[1][1]; }
      saved = last;
    }
    wg_sync<1>();
    const int si = slip == 1 ? 0 : 1, sq = slip == -1 ? 0 : 1;
    int i0 = raw[0][lane + si], i1 = raw[0][lane + 64 + si], q0 = raw[1][lane + sq], q1 = raw[1][lane + 64 + sq];
    if (detect) {
      /* decimation in time: bit-reversed load, then seven passes */
      re[__brev((unsigned)lane) >> 25] = (float)i0 / 32767.0f; im[__brev((unsigned)lane) >> 25] = (float)q0 / 32767.0f;
      re[__brev((unsigned)(lane + 64)) >> 25] = (float)i1 / 32767.0f; im[__brev((unsigned)(lane + 64)) >> 25] = (float)q1 / 32767.0f;
      wg_sync<1>();
      for (int half = 1; half < 128; half <<= 1) {
        const int k = lane & (half - 1), a = ((lane - k) << 1) + k, bb = a + half;
        const float wr = p.tw[2 * (k * (64 / half))], wi = p.tw[2 * (k * (64 / half)) + 1];
        const float xr = re[bb] * wr - im[bb] * wi, xi = re[bb] * wi + im[bb] * wr;
        const float ar = re[a], ai = im[a];
        wg_sync<1>();
        re[a] = ar + xr; im[a] = ai + xi; re[bb] = ar - xr; im[bb] = ai - xi;
        wg_sync<1>();
      }
      const float m0 = sqrtf(re[lane] * re[lane] + im[lane] * im[lane]), m1 = sqrtf(re[lane + 64] * re[lane + 64] + im[lane + 64] * im[lane + 64]);
      wg_sync<1>();
      re[lane] = m0; re[lane + 64] = m1;
      wg_sync<1>();
      if (lane == 0) {
        float top = 0.0f, sum = 0.0f;
        int at = 0;
        for (int k = 5; k < 123; k++) {
          sum = sum + re[k];
          if (re[k] > top) { top = re[k]; at = k; }
        }
        const float mean = sum / 118.0f;
        if ((double)top > (double)mean * 10.0) {
          if (top / re[128 - at] < 10.0f) {
            bad = (int16_t)(bad + 1);
            if (bad > 10) {
              int s = (int16_t)(slip + 1);
              bad = 0;
              if (s > 1) s = -1;
              slip = s;
              checks = 1;
            } else checks = (int16_t)(checks + 1);
          } else {
            checks = (int16_t)(checks + 1);
            bad = 0;
          }
        }
        if (checks > 1000) detect = 0;
        verdict[0] = slip; verdict[1] = bad; verdict[2] = checks; verdict[3] = detect;
      }
      wg_sync<1>();
      slip = verdict[0]; bad = verdict[1]; checks = verdict[2]; detect = verdict[3];
    }
    if (p.swap) { int t = i0; i0 = q0; q0 = t; t = i1; i1 = q1; q1 = t; }
    dst[(size_t)b * BS + lane] = (int)((unsigned)(i0 & 0xffff) | ((unsigned)q0 << 16));
    dst[(size_t)b * BS + lane + 64] = (int)((unsigned)(i1 & 0xffff) | ((unsigned)q1 << 16));
    wg_sync<1>();
  }
  if (lane == 0) { st[0] = (int16_t)slip; st[1] = (int16_t)saved; st[2] = (int16_t)bad; st[3] = (int16_t)checks; st[4] = (int16_t)detect; }
}
}  // namespace

struct rdsp_preproc {
  int n_channels, device, swap, restart;
  int16_t *d_st = nullptr;
  float *d_tw = nullptr;
};

extern "C" {
void rdsp_preproc_destroy(rdsp_preproc_t *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  if (p->d_st) (void)hipFree(p->d_st);
  if (p->d_tw) (void)hipFree(p->d_tw);
  delete p;
}
int rdsp_preproc_create(int n_channels, int device, rdsp_preproc_t **out) {
  if (!out || n_channels < 1) return RDSP_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
    rdsp_set_error("rdsp_preproc_create: no HIP device (this library has no CPU path)");
    return RDSP_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= count || hipSetDevice(device) != hipSuccess) return RDSP_ERR_INVALID;
  rdsp_preproc_t *p = new rdsp_preproc();
  p->n_channels = n_channels; p->device = device; p->swap = 0; p->restart = 0;
  float tw[128];
  for (int k = 0; k < 64; k++) { tw[2 * k] = (float)cos(-2.0 * 3.14159265358979323846 * k / 128.0); tw[2 * k + 1] = (float)sin(-2.0 * 3.14159265358979323846 * k / 128.0); }
  hipError_t err = hipMalloc((void **)&p->d_st, (size_t)n_channels * 6 * sizeof(int16_t));
  if (err == hipSuccess) err = hipMalloc((void **)&p->d_tw, sizeof tw);
  if (err == hipSuccess) err = hipMemset(p->d_st, 0, (size_t)n_channels * 6 * sizeof(int16_t)); /* the sketch's static initialiser: nothing detected, not detecting */
  if (err == hipSuccess) err = hipMemcpy(p->d_tw, tw, sizeof tw, hipMemcpyHostToDevice);
  if (err != hipSuccess) {
    rdsp_preproc_destroy(p);
    rdsp_set_error("rdsp_preproc_create: %s", hipGetErrorString(err));
    return RDSP_ERR_NOMEM;
  }
  *out = p;
  return RDSP_OK;
}
int rdsp_preproc_startAutoI2SerrorDetection(rdsp_preproc_t *p) { if (!p) return RDSP_ERR_INVALID; p->restart = 1; return RDSP_OK; } /* 0xf084 */
int rdsp_preproc_swapIQ(rdsp_preproc_t *p, int on) { if (!p) return RDSP_ERR_INVALID; p->swap = on ? 1 : 0; return RDSP_OK; }
int rdsp_preproc_channels(const rdsp_preproc_t *p) { return p ? p->n_channels : 0; }
int rdsp_preproc_device(const rdsp_preproc_t *p) { return p ? p->device : -1; }
/* AudioSDRpreProcessor::update() for n_blocks consecutive blocks of every channel; d_out may be d_iq */
int rdsp_preproc_update(rdsp_preproc_t *p, const int16_t *d_iq, size_t in_stride, int n_blocks, int16_t *d_out, size_t out_stride, void *stream) {
  if (!p || !d_iq || !d_out || n_blocks < 0 || in_stride < (size_t)n_blocks * BS || out_stride < (size_t)n_blocks * BS) return RDSP_ERR_INVALID;
  if (n_blocks == 0) return RDSP_OK;
  hipError_t err = hipSetDevice(p->device);
  if (err != hipSuccess) return engine_fail("rdsp_preproc_update", err);
  PreParams a;
  a.iq = (const int32_t *)d_iq; a.in_stride = in_stride; a.out = (int32_t *)d_out; a.out_stride = out_stride;
  a.n_channels = p->n_channels; a.n_blocks = n_blocks; a.swap = p->swap; a.restart = p->restart; a.st = p->d_st; a.tw = p->d_tw;
  hipLaunchKernelGGL(rdsp_preproc_kernel, dim3((unsigned)p->n_channels), dim3(64), 0, (hipStream_t)stream, a);
  err = hipGetLastError();
  if (err != hipSuccess) return engine_fail("rdsp_preproc_update launch", err);
  p->restart = 0;
  return RDSP_OK;
}
/* [n_channels][4]: remedy in force (0 none, 1 I one sample later, -1 Q one sample later), bad count, counted blocks, detecting */
int rdsp_preproc_get_state(rdsp_preproc_t *p, int16_t *host_out, void *stream) {
  if (!p || !host_out) return RDSP_ERR_INVALID;
  std::vector<int16_t> st((size_t)p->n_channels * 6);
  hipError_t err = hipSetDevice(p->device);
  if (err == hipSuccess) err = hipMemcpyAsync(st.data(), p->d_st, st.size() * 2, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (err == hipSuccess) err = hipStreamSynchronize((hipStream_t)stream);
  if (err != hipSuccess) return engine_fail("rdsp_preproc_get_state", err);
  for (int c = 0; c < p->n_channels; c++) {
    host_out[4 * c] = st[6 * (size_t)c]; host_out[4 * c + 1] = st[6 * (size_t)c + 2]; host_out[4 * c + 2] = st[6 * (size_t)c + 3]; host_out[4 * c + 3] = st[6 * (size_t)c + 4];
  }
  return RDSP_OK;
}
}  // extern "C"
