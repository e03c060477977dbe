/*
 * rdsp_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the per-block IQ
 * receive chain.  No MFMA: the path is streaming FIR/FFT work in fp32.
 *
 *   rdsp_front_kernel<N,P,DECIM>  one channel per workgroup of NT = N/P threads
 *       A1  int16 IQ unpack           RDSP_convolutional.h:241-242
 *       A2  NCO mixer                 (AudioSDR, build-defined)
 *       A3  256-tap polyphase /4 FIR  (build-defined)
 *       A5  overlap-save filter       RDSP_convolutional.h:256-318
 *       A6  spectral subtraction NR   backup/RDSP_convolutional_spec.h:182-238
 *       demod select, and when no NLMS stage is active: A9 AGC, output gain,
 *       A10 pack                      RDSP_convolutional.h:342-350
 *   rdsp_tail_kernel<LPC>         one channel per LPC lanes (serial-in-time)
 *       A7  NLMS noise reduction      RDSP_noise_reduction.h:35-80
 *       A8  ALS notch / peak          (AudioSDR, build-defined on A7's core)
 *       A9  AGC, output gain, A10 pack
 *
 * Data movement: int16 IQ is read once with 16-byte coalesced loads (prefetched
 * one chunk ahead), everything between stays in LDS/registers, and audio is
 * written once.  Per-channel state (FIR history, overlap block, NFloor, AGC
 * gain, NLMS weights) is read at launch start and written back at the end, so
 * its traffic is amortised over the time batch.
 */
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "rdsp_front.h"
#include "rdsp_wave.h"

using namespace rdsp;

namespace {

/* ---- front kernel -------------------------------------------------------- */
/* LDS plan of the front kernel (float2 units), shared with the launch code */
/* FMX: the decimating FIR as a GEMM with v_mfma (fir_matrix, rdsp_front.h) instead of packed FMAs */
template <int N, int P, int DECIM, bool FMX>
struct FrontLds {
  static constexpr int NT = N / P;
  static constexpr int H = N / 2;
  static constexpr bool FM = FMX && (DECIM == 4);
  static constexpr int XS_N = (DECIM == 4) ? (FM ? RDSP_XL_N : 16 * RDSP_XP) : 0;
  static constexpr int HB_N = (H > 256) ? H : 256; /* new samples of one chunk / one hop */
  /* one-wave kernels with a work buffer that fits behind the FIR history reuse the planes */
  static constexpr bool ALIAS = (DECIM == 4) && (NT == 64) && (N <= 512);
  static constexpr int WB_N = ALIAS ? 0 : FftPlan<N, P>::WB;
  static constexpr int TAPS_N = (DECIM == 4) ? (FM ? RDSP_HZ_N / 2 : 128) : 0;
  static constexpr size_t BYTES = (size_t)(XS_N + HB_N + WB_N + TAPS_N) * sizeof(float2) + 64 * sizeof(float);
};
/* arm_sin_f32 / arm_cos_f32 of CMSIS-DSP as published (FastMathFunctions): the angle in turns, its fractional
 * part times 512 as a table index, linear interpolation between neighbouring entries of the 513-entry table.
 * `in` = x * 0.159154943092f for the sine, + 0.25f for the cosine. */
__device__ __forceinline__ float arm_fast_sin_turns(float in, const float *tab) {
  int n = (int)in;
  if (in < 0.0f) n--;
  in = in - (float)n;
  float findex = 512.0f * in;
  int index = (int)findex;
  if (index >= 512) { index = 0; findex -= 512.0f; }
  const float fract = findex - (float)index;
  const auto gt = (const __attribute__((address_space(1))) float *)tab; /* a global load, not a FLAT one */
  const float a = gt[index], b = gt[index + 1];
  return (1.0f - fract) * a + fract * b;
}
/* SPEC:213-217 and 226-235 as written, for the P bins of a thread: the new magnitude (0.2 mag at or under the
 * floor, mag - floor above it) and the bin rebuilt from it and the original phase,
 *   phi = atan2(im, re);  re' = mag' arm_cos_f32(phi);  im' = mag' arm_sin_f32(phi).
 * An opt-in mode (rdsp_set_spectral_resynthesis) inside kernels whose register budget decides their occupancy: the
 * bins go through the transform's work buffer in LDS -- `slot(e)`: the thread's own entries of the last forward /
 * first inverse pass -- and ONE rolled loop does the work, so the mode costs the default path
 * no registers (unrolled in place, sixteen atan2 chains took the 512-point kernel from 176 to 253 VGPRs). */
template <int P, typename SLOT>
__device__ __forceinline__ void spec_resynthesize_literal(float2 (&v)[P], float floor_, const float *tab, float2 *wb, SLOT slot) {
#pragma unroll
  for (int e = 0; e < P; e++) wb[slot(e)] = v[e];
#pragma unroll 1
  for (int e = 0; e < P; e++) {
    const float2 x = lds_ld(&wb[slot(e)]);
    const float pw = fmaf(x.y, x.y, fmaf(x.x, x.x, 1e-30f)); /* the same |X| as the caller's (SPEC:182) */
    const float m0 = pw * __builtin_amdgcn_rsqf(pw);
    const float m1 = (m0 <= floor_) ? 0.2f * m0 : m0 - floor_;                     /* SPEC:213-217 */
    const float turns = atan2f(x.y, x.x) * 0.159154943092f;                        /* SPEC:229 */
    wb[slot(e)] = make_float2(m1 * arm_fast_sin_turns(turns + 0.25f, tab),         /* SPEC:231 */
                              m1 * arm_fast_sin_turns(turns, tab));                /* SPEC:232 */
  }
#pragma unroll
  for (int e = 0; e < P; e++) v[e] = lds_ld(&wb[slot(e)]); /* single ds_read_b64, like the transform's passes */
}

/* SPEC:229-232 as written -- re' = mag' arm_cos_f32(phi), im' = mag' arm_sin_f32(phi), phi = atan2(im, re) -- evaluated
 * in closed form.  arm_sin_f32 interpolates linearly in a 512-step table: between the nodes phi0 and phi0 + h
 * (h = 2 pi / 512) at the fraction f it returns (1 - f) sin(phi0) + f sin(phi0 + h) = A(f) sin(phi) + B(f) cos(phi) with
 * A = (1 - f) cos(f h) + f cos((1 - f) h) = 1 - (h^2 / 2) f (1 - f) + O(h^4) and |B| < 3e-8; the cosine (phi + a quarter
 * turn = 128 table steps exactly) meets the same f.  So the as-written bin is the exact one, X mag'/mag, times A(f): what
 * the table's interpolation costs, 1.9e-5 of the bin at most -- and this expression is within 5e-8 of the table's own
 * arithmetic (tests/test_host_logic.py evaluates both over the circle).  f (1 - f) is the same in every octant, so
 * f comes from atan(min / max) alone: a degree-11 odd polynomial in table steps (error 1.4e-4 of a step, 1e-8 of the
 * result), no branches, no table, 14 operations a bin where atan2f and two interpolated look-ups took 95. */
__device__ __forceinline__ float spec_table_factor(float2 x) {
  const float ax = fabsf(x.x), ay = fabsf(x.y);
  const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
  const float z = mn * __builtin_amdgcn_rcpf(mx), s = z * z;
  const float u = fmaf(s, fmaf(s, fmaf(s, fmaf(s, fmaf(s, -0.954960883f, 4.29009151f), -9.48728275f), 15.7710886f), -27.1045456f), 81.4854736f) * z;
  const float f = __builtin_amdgcn_fractf(u);
  return fmaf(fmaf(-f, f, f), -7.52982e-05f, 1.0f); /* (2 pi / 512)^2 / 2 */
}

/* A field of the kernel's parameter block (or of the channel's group record) read where it is used, not at kernel
 * entry.  The compiler loads every kernarg it will ever need in the prologue, and with more than a hundred scalar values
 * live across the frame loop it spills them to VGPR lanes: 79 spilled SGPRs and ~95 v_readlane reloads per decimator
 * frame in the K2 instance of rdsp_front_fd_kernel, every one an issue slot of the vector unit.  What only the call's
 * first frame, an option's own branch or the state write-back at the end needs comes through here instead: a scalar load
 * from the kernarg segment through a pointer the optimizer cannot identify with the one it loaded from at entry (the
 * parameter block is the kernels' only argument: offset 0 of the segment). */
template <typename T>
__device__ __forceinline__ T kernarg_late(unsigned off) {
  auto kp = (const __attribute__((address_space(4))) unsigned char *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(kp));
  return *reinterpret_cast<const __attribute__((address_space(4))) T *>(kp + off);
}
#define RDSP_LATE(field) kernarg_late<decltype(RdspFrontParams::field)>((unsigned)offsetof(RdspFrontParams, field))
/* the group record's cold fields (what the 256 history samples were mixed with): read like the record at kernel entry,
 * vector loads of a wave-uniform address, made scalar by v_readfirstlane */
__device__ __forceinline__ uint32_t group_late_word(uint32_t gi, unsigned off) {
  const uint32_t *gw = reinterpret_cast<const uint32_t *>(RDSP_LATE(groups) + gi) + off / 4;
  asm volatile("" : "+s"(gw));
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const __attribute__((address_space(1))) uint32_t *)gw);
}
__device__ __forceinline__ float2 group_late_f2(uint32_t gi, unsigned off) {
  return make_float2(__builtin_bit_cast(float, group_late_word(gi, off)), __builtin_bit_cast(float, group_late_word(gi, off + 4)));
}
#define RDSP_GROUP_LATE_F2(gi, field) group_late_f2(gi, (unsigned)offsetof(RdspGroup, field))
#define RDSP_GROUP_LATE_U32(gi, field) group_late_word(gi, (unsigned)offsetof(RdspGroup, field))

/* ---- A5/A6 + epilogue: one overlap-save frame of H = N/2 new samples ------------------
 * Shared by the front kernels (direct-form and FFT-domain decimator).  fetch(i) returns new
 * sample i of the hop from wherever the producer left it in LDS. */
template <int N, int P, bool WALIAS, typename TW, typename FETCH>
__device__ __forceinline__ void front_frame(const RdspFrontParams &p, const RdspGroup &G, const TW &tw,
                                            const LdsBases<N, P, WALIAS> &lb, float2 *wb, float *red,
                                            const float2 (&mreg)[P], uint32_t vadbits, float vad_inv,
                                            float2 (&vprev)[P / 2], float &nfloor, float &agc_g, float &am_dc,
                                            int &frame_idx, size_t ch, int tid, FETCH fetch) {
  using PL = FftPlan<N, P>;
  constexpr int NT = PL::NT;
  constexpr int NW = NT / 64;
  constexpr int H = N / 2;
  constexpr int PH = P / 2;
  constexpr int NB = H / RDSP_BLOCK; /* 128-blocks per hop */
  const int lane = tid & 63;
  const int wave = tid >> 6;
  {
  float2 v[P];
  /* CONV:267-285: [previous hop | current hop]; CONV:274-278: the current hop is
   * the next frame's previous hop (this thread's elements stay in its registers) */
#pragma unroll
  for (int j = 0; j < PH; j++) {
    v[j] = vprev[j];
    v[j + PH] = fetch(tid + j * NT);
    vprev[j] = v[j + PH];
  }
  auto sync = []() { wg_sync<NW>(); };
  {
    float2 twp[P - 1];
    tw.template get<0>(twp);
    fwd_pass0_store<N, P>(lb, v, wb, twp); /* CONV:291 */
  }
  wg_sync<NW>();
  fwd_mid_all<N, P, 1, PL::NP - 1, WALIAS>(lb, wb, tw, sync);
  fwd_pass_last<N, P>(lb, v, wb);

  if (p.spectral_on) { /* SPEC:182-235 on the un-masked spectrum */
    float mag[P], rmag[P];
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < P; e++) {
      /* |X| and 1/|X| from one v_rsq_f32 (1 ulp) instead of a correctly rounded sqrt and a
       * division per bin; the floor keeps rsq finite on empty bins (|X| = 1e-15 there; it is
       * absorbed by any power above 1e-22) */
      const float pw = fmaf(v[e].y, v[e].y, fmaf(v[e].x, v[e].x, 1e-30f)); /* the floor rides in the sum */
      rmag[e] = __builtin_amdgcn_rsqf(pw);
      mag[e] = pw * rmag[e];                              /* SPEC:182 */
      part += ((vadbits >> e) & 1u) ? mag[e] : 0.f;       /* SPEC:194-197 */
    }
    float tot = wave_sum(part);
    if constexpr (NW > 1) {
      if (lane == 0) red[wave] = tot;
      wg_sync<NW>();
      tot = (red[0] + red[1]) + (red[2] + red[3]);
      wg_sync<NW>();
    }
    float th = tot * vad_inv;                      /* SPEC:200 */
    th = th * p.spectral_k;                        /* SPEC:202 */
    if (p.spectral_on == 2) {
      nfloor = th;                                 /* BK_INO:1595-1596: no smoothing */
    } else {
      nfloor += (th - nfloor) * 0.65f;             /* SPEC:205 */
      nfloor = nfloor > 0.f ? nfloor : 0.f;        /* SPEC:206 */
    }
    if (p.spectral_literal == 1) { /* rdsp_set_spectral_resynthesis(c, 1): SPEC:213-217, 226-235 as written, the table's interpolation in closed form */
#pragma unroll
      for (int e = 0; e < P; e++) {
        const float sc = ((mag[e] <= nfloor) ? 0.2f : fmaf(-nfloor, rmag[e], 1.f)) * spec_table_factor(v[e]);
        v[e].x *= sc;
        v[e].y *= sc;
      }
    } else if (p.spectral_literal) { /* (c, 2): the same with atan2f and the table looked up */
      /* the thread's own P entries of the work buffer: what it read in the last forward pass and writes in the
       * first inverse pass, so no other lane ever touches them in between (and they are inside the buffer under
       * either map, also where it is cut into the FIR planes behind their history) */
      const int own = lb.bi[PL::NP - 1];
      spec_resynthesize_literal<P>(v, nfloor, RDSP_LATE(sin_table), wb, [&](int e) { return own + e; });
    } else {
#pragma unroll
      for (int e = 0; e < P; e++) {
        /* SPEC:213-217, 226-235: X * mag'/mag with mag' = 0.2 mag at or under the floor and
         * mag - floor above it, i.e. a gain of 0.2 or 1 - floor/mag (an empty bin stays 0) */
        const float sc = (mag[e] <= nfloor) ? 0.2f : fmaf(-nfloor, rmag[e], 1.f);
        v[e].x *= sc;
        v[e].y *= sc;
      }
    }
  }
  /* CONV:301: spectrum x mask */
#pragma unroll
  for (int e = 0; e < P; e++) v[e] = cmul(v[e], mreg[e]);

  inv_pass_last<N, P>(lb, v, wb); /* CONV:309 */
  wg_sync<NW>();
  inv_mid_all<N, P, PL::NP - 2, WALIAS>(lb, wb, tw, sync);
  {
    float2 twp[P - 1];
    tw.template get<0>(twp);
    inv_pass0_load<N, P>(lb, v, wb, twp);
  }
  wg_sync<NW>(); /* wb is free again (next frame / taps / FIR partials) */

  /* CONV:314-318: keep the second half.  v[PH + jj] = y[N/2 + tid + jj*NT] */
  float L[PH], R[PH];
#pragma unroll
  for (int jj = 0; jj < PH; jj++) {
    L[jj] = v[PH + jj].x;
    R[jj] = v[PH + jj].y;
  }

  /* helper: per-128-block sums of a per-thread value over the workgroup */
  float bs[NB];
  auto block_sums = [&](const float(&pv)[PH]) {
#pragma unroll
    for (int jj = 0; jj < PH; jj++) {
      float s = wave_sum(pv[jj]);
      if (lane == 0) red[wave * PH + jj] = s;
    }
    wg_sync<NW>();
#pragma unroll
    for (int b = 0; b < NB; b++) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; w++)
#pragma unroll
        for (int jj = 0; jj < PH; jj++)
          if (((jj * NT + w * 64) >> 7) == b) s += red[w * PH + jj];
      bs[b] = s;
    }
    wg_sync<NW>();
  };

  if (G.demod == RDSP_K_DEMOD_REAL) {
#pragma unroll
    for (int jj = 0; jj < PH; jj++) R[jj] = L[jj];
  } else if (G.demod == RDSP_K_DEMOD_AM) {
    float a[PH];
#pragma unroll
    for (int jj = 0; jj < PH; jj++) a[jj] = __builtin_amdgcn_sqrtf(L[jj] * L[jj] + R[jj] * R[jj]);
    block_sums(a);
    float d0[NB], d1[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
      float m = bs[b] / (float)RDSP_BLOCK;
      float dn = am_dc + 0.25f * (m - am_dc);
      d0[b] = am_dc;
      d1[b] = dn;
      am_dc = dn;
    }
#pragma unroll
    for (int jj = 0; jj < PH; jj++) {
      const int b0 = (jj * NT) >> 7;
      float s0 = d0[b0], s1 = d1[b0];
      if constexpr (NT == 256) {
        if (tid >= 128) { s0 = d0[b0 + 1]; s1 = d1[b0 + 1]; }
      }
      int i = (tid + jj * NT) & 127;
      float dc = s0 + (s1 - s0) * ((float)(i + 1) / (float)RDSP_BLOCK);
      L[jj] = a[jj] - dc;
      R[jj] = L[jj];
    }
  }

  const size_t tout = (size_t)frame_idx * H;
  if (p.to_mid) {
#pragma unroll
    for (int jj = 0; jj < PH; jj++) p.mid[ch * p.mid_stride + tout + tid + jj * NT] = L[jj];
    if (G.demod == RDSP_K_DEMOD_SAM) { /* the PLL stage needs the quadrature part too */
#pragma unroll
      for (int jj = 0; jj < PH; jj++) p.mid_q[ch * p.mid_stride + tout + tid + jj * NT] = R[jj];
    }
  } else {
    if (p.agc_on) {
      float pw[PH];
#pragma unroll
      for (int jj = 0; jj < PH; jj++) pw[jj] = L[jj] * L[jj] + R[jj] * R[jj];
      block_sums(pw);
      float g0[NB], g1[NB];
#pragma unroll
      for (int b = 0; b < NB; b++) {
        float pp = bs[b] / (float)(2 * RDSP_BLOCK);
        float rms = __builtin_amdgcn_sqrtf(pp); /* 1 ulp; the loop gain is a contraction */
        float gt = 0.25f * __builtin_amdgcn_rcpf(rms + 1e-6f);
        gt = fminf(gt, 100.0f);
        float coef = (gt < agc_g) ? p.agc_attack : p.agc_decay;
        float gn = agc_g + coef * (gt - agc_g);
        g0[b] = agc_g;
        g1[b] = gn;
        agc_g = gn;
      }
#pragma unroll
      for (int jj = 0; jj < PH; jj++) {
        const int b0 = (jj * NT) >> 7;
        float s0 = g0[b0], s1 = g1[b0];
        if constexpr (NT == 256) {
          if (tid >= 128) { s0 = g0[b0 + 1]; s1 = g1[b0 + 1]; }
        }
        int i = (tid + jj * NT) & 127;
        float g = s0 + (s1 - s0) * ((float)(i + 1) / (float)RDSP_BLOCK);
        L[jj] *= g;
        R[jj] *= g;
      }
    }
#pragma unroll
    for (int jj = 0; jj < PH; jj++) {
      float l = L[jj] * p.out_gain, r = R[jj] * p.out_gain;
      size_t o = ch * p.out_stride + tout + tid + jj * NT;
      __builtin_nontemporal_store(pack_lr(l, r), p.out_i16 + o); /* CONV:346-347; written once, read by nobody here */
      if (p.out_f32) p.out_f32[o] = make_float2(l, r);
    }
  }
  frame_idx++;
  }
}

/* ---- FFT_L = 256 behind the frequency-domain decimator: FOUR overlap-save frames per pass ----------
 * 256 points over a whole wave are 4 points per lane: four radix-4 passes, three LDS exchanges each way,
 * every pass a quarter-filled instruction stream -- the filter stage of K1 / K2 (the reference's own
 * FFT_L, CONV:36) cost as many VALU instructions and more LDS cycles than the 256-tap decimator in front
 * of it.  Overlap-save frames do not depend on each other (each is two consecutive hops of the decimated
 * stream, which sits in the ring), so here a 16-lane DPP row takes one frame -- 16 points per lane, two
 * radix-16 passes, ONE exchange each way -- and the wave takes four consecutive frames at once: a third
 * of the LDS operations per frame and about half the instructions.  What IS sequential across frames
 * (NFloor SPEC:205, the AGC gain, the AM detector's DC) depends on one number per frame: the four row
 * sums are read out with v_readlane and the four steps of the recursion run on wave-uniform values.
 * The mask is read from the same device image as the radix-4 plan's (digit-reversed for FftPlan<256, 4>):
 * bin k = i + 16 e of lane i sits at 64 e1 + e0 + 16 i0 + 4 i1 (i = i0 + 4 i1, e = e0 + 4 e1).
 * Ring: eight hops of 128, each padded by 16 float2 so that the two rows of a 32-lane group read disjoint
 * halves of the 64 banks.  The hop in front of the oldest unconsumed one is never overwritten (it is the
 * first frame's overlap, CONV:267-271: no previous-hop register file as in front_frame): a decimator frame adds
 * 448 samples when at most 448 are unconsumed (a quad goes as soon as 512 are there, and the counts are
 * multiples of 64), 128 + 448 + 448 = the ring.
 * Cost of the shape: 226 VGPRs and 18 KiB of LDS per channel (the four rows' exchange buffers), where the
 * one-frame form takes 187 and 12.9.  Alone that is still two waves per SIMD and eight channels per CU, and
 * K2 runs 0.6775 -> 0.6115 ms per step (same-box A/B); beside a tail kernel (124 VGPRs, 12.5 KiB per four
 * channels) it would be one wave per SIMD, so chains that hand their audio to the tail kernel keep the
 * one-frame form (template Q4, chosen by the launch code).  Moving mask and twiddles to LDS instead
 * (171 VGPRs, 22-23 KiB: seven or six channels per CU) measured 0.710 / 0.744 ms: occupancy is worth more. */
constexpr int QUAD_HOPS = 8, QUAD_PITCH = 128 + 16, QUAD_RING = QUAD_HOPS * QUAD_PITCH;
constexpr int QUAD_WB = 4 * FftPlan<256, 16>::WB;

template <typename F>
__device__ __forceinline__ void quad_chain(float x, int g, int nf, float &state, float &before, float &after, F step) {
  const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
  const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
  const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
  const float x3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
  const float s0 = state, s1 = step(s0, x0), s2 = step(s1, x1), s3 = step(s2, x2), s4 = step(s3, x3);
  before = g == 0 ? s0 : (g == 1 ? s1 : (g == 2 ? s2 : s3));
  after = g == 0 ? s1 : (g == 1 ? s2 : (g == 2 ? s3 : s4));
  state = nf == 1 ? s1 : (nf == 2 ? s2 : (nf == 3 ? s3 : s4)); /* frames g >= nf are not there: their sums are never used */
}

template <int HOPS = QUAD_HOPS, typename TW>
__device__ __forceinline__ void front_frame_quad(const RdspFrontParams &p, const RdspGroup &G, const TW &tw,
                                                 const LdsBases<256, 16, false> &lb, float2 *wbg, const float2 *ring,
                                                 int rhop, int nf, int mbase, uint32_t vadbits, float vad_inv, float &nfloor,
                                                 float &agc_g, float &am_dc, int frame_idx, size_t ch, int lane) {
  constexpr int N = 256, P = 16;
  const int g = lane >> 4, i = lane & 15;
  int hc = rhop + g;
  hc = hc >= HOPS ? hc - HOPS : hc;
  const int hp = hc == 0 ? HOPS - 1 : hc - 1;
  const float2 *cur = ring + hc * QUAD_PITCH + i, *prv = ring + hp * QUAD_PITCH + i;
  /* this lane's sixteen bins of the group's mask: L2-resident, land behind the forward transform */
  float2 mreg[P];
  {
    const float2 *mp = p.mask_pool + G.mask_off;
    asm volatile("" : "+s"(mp));
    const auto gp = as_global(mp);
#pragma unroll
    for (int e = 0; e < P; e++) mreg[e] = gp[mbase + 64 * (e >> 2) + (e & 3)];
  }
  float2 v[P];
#pragma unroll
  for (int j = 0; j < P / 2; j++) { /* CONV:267-285: [previous hop | current hop], v[j] = x[i + 16 j] */
    v[j] = lds_ld(prv + 16 * j);
    v[j + P / 2] = lds_ld(cur + 16 * j);
  }
  {
    float2 twp[P - 1];
    tw.template get<0>(twp);
    fwd_pass0_store<N, P>(lb, v, wbg, twp); /* CONV:291 */
  }
  wg_sync<1>();
  fwd_pass_last<N, P>(lb, v, wbg);

  if (p.spectral_on) { /* SPEC:182-235 on the un-masked spectrum, as front_frame */
    float mag[P], rmag[P];
    float part = 0.f;
#pragma unroll
    for (int e = 0; e < P; e++) {
      const float pw = fmaf(v[e].y, v[e].y, fmaf(v[e].x, v[e].x, 1e-30f));
      rmag[e] = __builtin_amdgcn_rsqf(pw);
      mag[e] = pw * rmag[e];                              /* SPEC:182 */
      part += ((vadbits >> e) & 1u) ? mag[e] : 0.f;       /* SPEC:194-197 */
    }
    float th = row_allsum(part) * vad_inv;                /* SPEC:200 */
    th = th * p.spectral_k;                               /* SPEC:202 */
    float nf0, mine;
    const int old_variant = p.spectral_on == 2;
    quad_chain(th, g, nf, nfloor, nf0, mine, [&](float s, float t) {
      float n = s + (t - s) * 0.65f;                      /* SPEC:205 */
      n = n > 0.f ? n : 0.f;                              /* SPEC:206 */
      return old_variant ? t : n;                         /* BK_INO:1595-1596: no smoothing */
    });
    if (p.spectral_literal == 1) { /* SPEC:226-235 as written, as in front_frame */
#pragma unroll
      for (int e = 0; e < P; e++) {
        const float sc = ((mag[e] <= mine) ? 0.2f : fmaf(-mine, rmag[e], 1.f)) * spec_table_factor(v[e]);
        v[e].x *= sc;
        v[e].y *= sc;
      }
    } else if (p.spectral_literal) { /* the same with atan2f and the table looked up */
      const int own = lb.bi[FftPlan<256, 16>::NP - 1];
      spec_resynthesize_literal<P>(v, mine, RDSP_LATE(sin_table), wbg, [&](int e) { return own + e; });
    } else {
#pragma unroll
      for (int e = 0; e < P; e++) {
        const float sc = (mag[e] <= mine) ? 0.2f : fmaf(-mine, rmag[e], 1.f); /* SPEC:213-217, 226-235 */
        v[e].x *= sc;
        v[e].y *= sc;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < P; e++) v[e] = cmul(v[e], mreg[e]); /* CONV:301 */

  inv_pass_last<N, P>(lb, v, wbg); /* CONV:309 */
  wg_sync<1>();
  {
    float2 twp[P - 1];
    tw.template get<0>(twp);
    inv_pass0_load<N, P>(lb, v, wbg, twp);
  }
  wg_sync<1>();

  /* CONV:314-318: keep the second half.  v[8 + j] = y[128 + i + 16 j]; sample o = i + 16 j of the hop */
  constexpr int Q = P / 2;
  float L[Q], R[Q], ramp[Q];
#pragma unroll
  for (int j = 0; j < Q; j++) {
    L[j] = v[Q + j].x;
    R[j] = v[Q + j].y;
    ramp[j] = (float)(i + 16 * j + 1) / (float)RDSP_BLOCK;
  }
  if (G.demod == RDSP_K_DEMOD_REAL) {
#pragma unroll
    for (int j = 0; j < Q; j++) R[j] = L[j];
  } else if (G.demod == RDSP_K_DEMOD_AM) {
    float a[Q], s = 0.f;
#pragma unroll
    for (int j = 0; j < Q; j++) {
      a[j] = __builtin_amdgcn_sqrtf(L[j] * L[j] + R[j] * R[j]);
      s += a[j];
    }
    float d0, d1;
    quad_chain(row_allsum(s), g, nf, am_dc, d0, d1, [&](float dc, float sum) {
      const float m = sum / (float)RDSP_BLOCK;
      return dc + 0.25f * (m - dc);
    });
#pragma unroll
    for (int j = 0; j < Q; j++) {
      L[j] = a[j] - (d0 + (d1 - d0) * ramp[j]);
      R[j] = L[j];
    }
  }
  const bool valid = g < nf;
  const size_t tout = (size_t)(frame_idx + g) * RDSP_BLOCK + (size_t)i;
  { /* the launch code takes this form only for chains whose audio ends here (no intermediate for a tail stage) */
    if (p.agc_on) {
      float pw = 0.f;
#pragma unroll
      for (int j = 0; j < Q; j++) pw += L[j] * L[j] + R[j] * R[j];
      float g0, g1;
      quad_chain(row_allsum(pw), g, nf, agc_g, g0, g1, [&](float gain, float sum) {
        const float pp = sum / (float)(2 * RDSP_BLOCK);
        const float rms = __builtin_amdgcn_sqrtf(pp);
        float gt = 0.25f * __builtin_amdgcn_rcpf(rms + 1e-6f);
        gt = fminf(gt, 100.0f);
        const float coef = (gt < gain) ? p.agc_attack : p.agc_decay;
        return gain + coef * (gt - gain);
      });
#pragma unroll
      for (int j = 0; j < Q; j++) {
        const float gg = g0 + (g1 - g0) * ramp[j];
        L[j] *= gg;
        R[j] *= gg;
      }
    }
    if (valid) {
#pragma unroll
      for (int j = 0; j < Q; j++) {
        const float l = L[j] * p.out_gain, r = R[j] * p.out_gain;
        const size_t o = ch * p.out_stride + tout + 16 * j;
        __builtin_nontemporal_store(pack_lr(l, r), p.out_i16 + o); /* CONV:346-347 */
        if (p.out_f32) p.out_f32[o] = make_float2(l, r);
      }
    }
  }
}

/* LEAN = true trades registers for a little recomputation (twiddle powers per pass,
 * mask slice re-read per chunk).  It pays at radix 16, where it buys the second wave per
 * SIMD.  At radix 8 it was what let two front waves and a tail wave share the 512-register
 * file of a SIMD in pipelined mode; since the butterflies and the FIR were written out by
 * hand the full-register kernel needs 185 VGPRs, fits as well (2 x 192 + 112) and is the
 * default in both modes (the lean one stays selectable, rdsp_chain_set_front_variant).
 * FMX = true (opt-in, rdsp_chain_set_fir_variant) runs the decimating FIR as
 * v_mfma_f32_16x16x4_f32 GEMM slices.  fp32 MFMA and fp32 VALU work do not overlap on a gfx950
 * SIMD (tests/micro/mfma_valu_overlap.hip: one wave of each takes the sum of both times), so
 * this is not a second pipe; it wins 10 % at K2, 6 % on the K3 front kernel and 2 % at K4
 * through fewer LDS reads and instructions and 40-60 fewer VGPRs -- and its 32-cycle
 * instructions starve a co-resident tail wave (pipelined K3: 2.21 -> 2.58 ms). */
template <int N, int P, int DECIM, bool LEAN, bool PRE, bool FMX>
__global__ void __launch_bounds__(N / P, 2) rdsp_front_kernel(RdspFrontParams p) {
  using PL = FftPlan<N, P>;
  constexpr int NT = PL::NT;
  constexpr int NW = NT / 64;
  constexpr int H = N / 2;
  constexpr int PH = P / 2;
  constexpr int CH_OUT = 256;
  constexpr int CH_IN = CH_OUT * DECIM;
  constexpr int FPC = (H >= CH_OUT) ? 1 : CH_OUT / H; /* frames per chunk */
  constexpr int CPF = (H >= CH_OUT) ? H / CH_OUT : 1; /* chunks per frame */
  using LY = FrontLds<N, P, DECIM, FMX>;
  constexpr bool ALIAS = LY::ALIAS;
  constexpr bool FM = LY::FM;
  /* matrix FIR: the input is one padded line; after the FIR only its first 256 samples (the
   * history) are live, so the work buffer sits right behind them with the plain map */
  constexpr bool WALIAS = ALIAS && !FM;
  constexpr int LP = (CH_IN / 4 + NT - 1) / NT; /* uint4 loads per thread per chunk */
  static_assert(DECIM == 1 || DECIM == 4, "decimation 1 or 4");
  static_assert(NT == 64 || NT == 256, "one or four waves per channel");
  static_assert(NW == 1 || PL::WB >= 4 * CH_OUT, "work buffer holds the FIR partial sums");
  static_assert(LP == 4 || LP == 1, "the history phasor below follows the scatter loop's passes");

  /* LDS: [polyphase planes | new hop(s) | work buffer (unless aliased into the
   * planes) | decimator taps | reduction scratch].  The previous hop is not in
   * LDS: every thread keeps its own P/2 elements of it in registers (vprev). */
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *xs = reinterpret_cast<float2 *>(smem_raw);
  float2 *hb = xs + LY::XS_N;
  float2 *wb = ALIAS ? (FM ? xs + xl_pos(0) : xs) : hb + LY::HB_N;
  static_assert(!FM || !ALIAS || xl_pos(0) + PL::WB <= RDSP_XL_N, "work buffer fits behind the history");
  float4 *taps_lds = reinterpret_cast<float4 *>(hb + LY::HB_N + (ALIAS ? 0 : PL::WB));
  float *hz = reinterpret_cast<float *>(taps_lds);
  float *red = reinterpret_cast<float *>(reinterpret_cast<float2 *>(taps_lds) + LY::TAPS_N);

  /* PRE: the pre-processor's IQ swap and the noise blanker are compiled in (their
   * run-time tests inside the unpack loop cost ~2 % when both are off, measured) */
  const bool NB_ON = PRE && p.nb_on != 0, SWAP_IQ = PRE && p.swap_iq != 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const size_t ch = (size_t)p.ch_base + blockIdx.x;
  const uint32_t *iq = p.iq + ch * p.in_stride;
  /* this channel's group record into scalar registers */
  RdspGroup G;
  {
    const uint32_t gi = p.group_of ? (uint32_t)p.group_of[ch] : 0u;
    const uint32_t *gw = reinterpret_cast<const uint32_t *>(p.groups + gi);
    uint32_t r[32];
#pragma unroll
    for (int i = 0; i < 32; i++) r[i] = (i < 24) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)gw[i]) : 0u;
    G = __builtin_bit_cast(RdspGroup, r);
  }

  /* first uint4 loads of chunk 0 go out before anything else */
  uint4 raw[LP];
#pragma unroll
  for (int k = 0; k < LP; k++) {
    int idx = tid + NT * k;
    if (idx < CH_IN / 4) raw[k] = *reinterpret_cast<const uint4 *>(iq + 4 * idx);
  }

  /* per-thread constants that stay in registers for the whole launch: FFT
   * twiddles, LDS bases of every pass, this thread's slice of the filter mask
   * (digit-reversed, /N), its VAD-bin membership bits and its four taps */
  Twiddles<N, P, LEAN> tw;
  tw.init(tid);
  LdsBases<N, P, WALIAS> lb;
  make_lds_bases<N, P, WALIAS>(tid, lb);
  uint32_t vadbits = 0;
#pragma unroll
  for (int e = 0; e < P; e++) {
    int k = bin_of_pos<N, P>(tid * P + e);
    if (k >= p.vad_lo && k <= p.vad_hi) vadbits |= 1u << e;
  }
  if constexpr (DECIM == 4) {
    if constexpr (FM) { /* tap line hz[t + 64] = h[t] = hc[t % 4][t / 4], zero outside 0..255 */
      for (int t = tid; t < RDSP_HZ_N; t += NT) {
        const int tt = t - 64;
        hz[t] = (tt >= 0 && tt < 256) ? p.fir_hc[(tt & 3) * 64 + (tt >> 2)] : 0.f;
      }
    } else {
      if (tid < 64) taps_lds[tid] = reinterpret_cast<const float4 *>(p.fir_hc)[tid];
    }
  }

  float nfloor = p.st_scal[ch * 4 + 0];
  const float vad_inv = 1.0f / (float)(p.vad_hi - p.vad_lo); /* SPEC:200, once per launch */
  float agc_g = p.st_scal[ch * 4 + 1];
  float am_dc = p.st_scal[ch * 4 + 2];
  float nb_level = p.st_scal[ch * 4 + 3];

  /* state in: previous hop -> registers, FIR history -> polyphase planes */
  float2 vprev[PH];
#pragma unroll
  for (int j = 0; j < PH; j++) vprev[j] = p.st_prev[ch * H + tid + j * NT];
  if constexpr (DECIM == 4) {
    for (int i = tid; i < 64; i += NT) {
      uint4 w4 = *reinterpret_cast<const uint4 *>(p.st_hist + ch * 256 + 4 * i);
      uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
      if (PRE && p.swap_hist != 0) { /* the stored history is the raw stream: swapped as the call it came in with swapped */
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = __builtin_amdgcn_alignbit(w[k], w[k], 16);
      }
      /* the same phasor arithmetic these samples went through as the last 256 of the
       * previous chunk (pass LP-1 of the scatter loop below), so that a stream gives
       * the same bits however it is cut into calls */
      float2 ph0 = make_float2(1.f, 0.f);
      if (G.dphi_hist != 0u) {
        if constexpr (LP == 4) {
          ph0 = nco_phasor_alu((p.n0 - (uint32_t)CH_IN + 4u * (uint32_t)i) * G.dphi_hist);
          ph0 = cmul_pinned_u(ph0, G.rothp3);
        } else {
          ph0 = nco_phasor_alu((p.n0 - 256u + 4u * (uint32_t)i) * G.dphi_hist);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float2 x = unpack_iq(w[k], PRE ? p.scale_i_hist : p.scale_i, PRE ? p.scale_q_hist : p.scale_q); /* ... and the gains of that call */
        { /* the history keeps the mixing it went through when it was new (identity phasors when
           * the NCO was off: roth* are (1, -0) then and the products are exact) */
          float2 ph = (k == 0) ? ph0 : cmul_pinned_u(ph0, k == 1 ? G.roth1 : (k == 2 ? G.roth2 : G.roth3));
          x = cmul_pinned(x, ph);
        }
        xs[FM ? xl_pos(-256 + 4 * i + k) : xs_pos(-256 + 4 * i + k)] = x;
      }
    }
  }
  int frame_idx = 0;
  wg_sync<NW>();

  for (int chunk = 0; chunk < p.n_chunks; chunk++) {
    /* ---- A1 + A2: unpack, gains, mix; scatter into the polyphase planes ----
     * One accurate phasor per thread per chunk (ALU only: no memory traffic in
     * the loop besides the IQ stream); the other samples of the thread follow
     * by constant rotations (k*4*NT samples between passes, 1..3 inside one). */
    /* this thread's slice of the mask (digit-reversed, /N, thread-major): L2-resident,
     * requested at the top of the chunk and consumed after the forward transform, so
     * its latency hides behind the FIR.  The pointer is made opaque so the loads are
     * not hoisted out of the chunk loop into 2P persistent registers. */
    float2 mreg[P];
    {
      const float2 *mp = p.mask_pool + G.mask_off;
      if constexpr (LEAN) asm volatile("" : "+s"(mp));
      const auto gp = as_global(mp);
#pragma unroll
      for (int e = 0; e < P; e++) mreg[e] = gp[e * NT + tid];
    }
    float2 ph_base = make_float2(1.f, 0.f);
    if (G.dphi != 0u)
      ph_base = nco_phasor_alu((p.n0 + (uint32_t)chunk * CH_IN + 4u * (uint32_t)tid) * G.dphi);
    /* noise blanker (engine feature, build-defined): one decision window per chunk; the
     * threshold comes from the windows before this one, so the chunk stays parallel */
    const float nb_t = nb_level * p.nb_thr;
    float nb_acc = 0.f;
#pragma unroll
    for (int k = 0; k < LP; k++) {
      int idx = tid + NT * k;
      if (idx < CH_IN / 4) {
        uint32_t w[4] = {raw[k].x, raw[k].y, raw[k].z, raw[k].w};
        if (SWAP_IQ) { /* preProcessor.swapIQ(true), INO:118 */
#pragma unroll
          for (int j = 0; j < 4; j++) w[j] = __builtin_amdgcn_alignbit(w[j], w[j], 16);
        }
        float2 ph0 = ph_base;
        if (k > 0) ph0 = cmul_pinned_u(ph_base, k == 1 ? G.rotp1 : (k == 2 ? G.rotp2 : G.rotp3));
        bool blanked[4] = {false, false, false, false};
        /* the four phasors first, then the four products: independent chains the scheduler can
         * interleave (each complex product is a dependent pair of packed instructions).
         * NCO off: the record's rotations are (1, -0) and every product is exact, so the
         * multiplies stay unconditional (a select per sample cost more than they do). */
        float2 ph[4], x[4];
        ph[0] = ph0;
        ph[1] = cmul_pinned_u(ph0, G.rot1);
        ph[2] = cmul_pinned_u(ph0, G.rot2);
        ph[3] = cmul_pinned_u(ph0, G.rot3);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          x[j] = unpack_iq(w[j], p.scale_i, p.scale_q);
          if (NB_ON) {
            const float pw = x[j].x * x[j].x + x[j].y * x[j].y;
            blanked[j] = nb_level > 0.f && pw > nb_t;
            x[j] = blanked[j] ? make_float2(0.f, 0.f) : x[j];
            nb_acc += blanked[j] ? 0.f : pw;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = cmul_pinned(x[j], ph[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if constexpr (DECIM == 4) {
            xs[FM ? xl_pos(4 * idx + j) : xs_pos(4 * idx + j)] = x[j];
          } else {
            int m = 4 * idx + j; /* no decimator: the sample is the "output" */
            hb[(chunk % CPF) * CH_OUT + m] = x[j];
          }
        }
        if (NB_ON) { /* a blanked sample stays blanked when it becomes FIR history */
          raw[k].x = blanked[0] ? 0u : raw[k].x;
          raw[k].y = blanked[1] ? 0u : raw[k].y;
          raw[k].z = blanked[2] ? 0u : raw[k].z;
          raw[k].w = blanked[3] ? 0u : raw[k].w;
        }
      }
    }
    if (NB_ON) {
      float tot = wave_sum(nb_acc);
      if constexpr (NW > 1) {
        if (lane == 0) red[wave] = tot;
        wg_sync<NW>();
        tot = (red[0] + red[1]) + (red[2] + red[3]);
        wg_sync<NW>();
      }
      const float mean = tot / (float)CH_IN;
      nb_level = (nb_level > 0.f) ? nb_level + 0.2f * (mean - nb_level) : mean;
    }
    /* prefetch the next chunk's raw samples; they land during FIR + FFT */
    if (chunk + 1 < p.n_chunks) {
#pragma unroll
      for (int k = 0; k < LP; k++) {
        int idx = tid + NT * k;
        if (idx < CH_IN / 4)
          raw[k] = *reinterpret_cast<const uint4 *>(iq + (size_t)(chunk + 1) * CH_IN + 4 * idx);
      }
    }
    wg_sync<NW>();

    /* ---- A3: polyphase decimating FIR ------------------------------------ */
    if constexpr (DECIM == 4 && FM) {
      /* on the matrix pipe (rdsp_front.h): lane 16 kq + i gets outputs m = 64 kq + 16 r + i */
      rdsp_v4f dre = {0.f, 0.f, 0.f, 0.f}, dim = {0.f, 0.f, 0.f, 0.f};
      const int mi = lane & 15, mk = lane >> 4;
      if constexpr (NW == 1) {
        if (p.front_prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (p.front_prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (p.front_prio == 3) __builtin_amdgcn_s_setprio(3);
        fir_matrix<0, 80>(lane, xs, hz, dre, dim);
        if (p.front_prio > 0) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int r = 0; r < 4; r++) hb[(chunk % CPF) * CH_OUT + 64 * mk + 16 * r + mi] = make_float2(dre[r], dim[r]);
        wg_sync<NW>();
      } else {
        /* four waves: wave w takes a quarter of the K-slices; partials summed via LDS */
        if (wave == 0) fir_matrix<0, 20>(lane, xs, hz, dre, dim);
        else if (wave == 1) fir_matrix<20, 40>(lane, xs, hz, dre, dim);
        else if (wave == 2) fir_matrix<40, 60>(lane, xs, hz, dre, dim);
        else fir_matrix<60, 80>(lane, xs, hz, dre, dim);
#pragma unroll
        for (int r = 0; r < 4; r++) wb[wave * CH_OUT + 64 * mk + 16 * r + mi] = make_float2(dre[r], dim[r]);
        wg_sync<NW>();
        {
          float2 s0 = wb[tid], s1 = wb[CH_OUT + tid], s2 = wb[2 * CH_OUT + tid], s3 = wb[3 * CH_OUT + tid];
          float2 s = cadd(cadd(s0, s1), cadd(s2, s3));
          hb[(chunk % CPF) * CH_OUT + tid] = s;
        }
      }
      /* the last 256 samples of the chunk are the next chunk's history */
      for (int t = tid; t < 256; t += NT) xs[xl_pos(t - 256)] = xs[xl_pos(768 + t)];
      wg_sync<NW>();
    } else if constexpr (DECIM == 4) {
      float2 acc[4];
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] = make_float2(0.f, 0.f);
      if constexpr (NW == 1) {
        /* when a tail-kernel wave shares the SIMD (pipelined mode), the FIR -- the
         * throughput-bound part -- takes issue priority; the rest of the chunk runs at
         * normal priority so the latency-bound tail keeps pace (measured balance) */
        if (p.front_prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (p.front_prio == 2) __builtin_amdgcn_s_setprio(2);
        else if (p.front_prio == 3) __builtin_amdgcn_s_setprio(3);
        fir_lane<(P >= 8)>(lane, 0, 4, xs, taps_lds, acc);
        if (p.front_prio > 0) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int r = 0; r < 4; r++) hb[(chunk % CPF) * CH_OUT + 4 * lane + r] = acc[r];
        wg_sync<NW>();
      } else {
        /* four waves: wave w takes polyphase branch w; partials summed via LDS */
        fir_lane<(P >= 8)>(lane, wave, wave + 1, xs, taps_lds, acc);
#pragma unroll
        for (int r = 0; r < 4; r++) wb[wave * CH_OUT + 4 * lane + r] = acc[r];
        wg_sync<NW>();
        {
          float2 s0 = wb[tid], s1 = wb[CH_OUT + tid], s2 = wb[2 * CH_OUT + tid], s3 = wb[3 * CH_OUT + tid];
          float2 s = cadd(cadd(s0, s1), cadd(s2, s3));
          hb[(chunk % CPF) * CH_OUT + tid] = s;
        }
      }
      /* slide the FIR history: entries 64..80 of every plane -> 0..16 */
      {
        float4 *xs4 = reinterpret_cast<float4 *>(xs);
        for (int i = tid; i < 8 * 17; i += NT) {
          int sp = i / 17, e = i % 17;
          xs4[sp * RDSP_XP + e] = xs4[sp * RDSP_XP + 64 + e];
        }
      }
      wg_sync<NW>();
    }

    if ((chunk + 1) % CPF != 0) continue;

    /* ---- A5/A6: overlap-save frames ---------------------------------------- */
#pragma unroll 1
    for (int f = 0; f < FPC; f++) {
      const float2 *hnew = hb + f * H;
      front_frame<N, P, WALIAS>(p, G, tw, lb, wb, red, mreg, vadbits, vad_inv, vprev, nfloor, agc_g, am_dc, frame_idx,
                                ch, tid, [&](int i) { return hnew[i]; }); /* advances frame_idx */
    }
  }

  /* ---- state out --------------------------------------------------------- */
#pragma unroll
  for (int j = 0; j < PH; j++) p.st_prev[ch * H + tid + j * NT] = vprev[j];
  if constexpr (DECIM == 4) {
    /* the last 256 input samples as they entered the FIR (blanked ones as zero): the
     * registers of the last load pass still hold them, no prefetch followed */
    if constexpr (LP == 4) {
      *reinterpret_cast<uint4 *>(p.st_hist + ch * 256 + 4 * tid) = raw[LP - 1];
    } else {
      if (tid >= NT - 64) *reinterpret_cast<uint4 *>(p.st_hist + ch * 256 + 4 * (tid - (NT - 64))) = raw[0];
    }
  } else {
    /* no FIR history at decim 1, but the call's last raw word still has a reader: the I2S slip correction,
     * switched on between two calls, pairs the next call's first sample with it (rdsp_chain_process) */
    if (tid == 0) p.st_hist[ch * 256 + 255] = iq[(size_t)p.n_chunks * CH_IN - 1];
  }
  if (tid == 0) {
    p.st_scal[ch * 4 + 0] = nfloor;
    if (!p.to_mid) p.st_scal[ch * 4 + 1] = agc_g;
    p.st_scal[ch * 4 + 2] = am_dc;
    p.st_scal[ch * 4 + 3] = nb_level;
  }
}

/* ---- front kernel with the decimator in the frequency domain -----------------------------
 * Same chain as rdsp_front_kernel<N, P, 4, ...>; stage A3 (y[m] = sum_{k<256} h[k] x[4m - k]) is
 * evaluated as a polyphase overlap-save convolution instead of 1024 packed FMAs per chunk and lane:
 *     x[4q + r] = X_r[q]  (r = 0..3: the four int16 pairs of one aligned 16-byte load),
 *     y[m] = sum_r sum_{k<=64} g_r[k] X_r[m - k],   g_r[k] = h[4k - r]  (zero outside 0..255),
 * i.e. four 512-point forward transforms of the mixed input at the LOW rate (512 whatever FFT_L is:
 * 448 of 512 outputs are valid, and the radix-8 passes are the cheapest per point), a
 * multiply-accumulate with the branch spectra G_r (host-computed, /512, digit-reversed like the
 * filter mask) and ONE inverse transform: 448 valid outputs per frame.  That is the 4N-point overlap-save decimator
 * with its first two radix-2 levels folded into the masks (only N of the 4N bins survive the
 * fold by 4).  Per frame and lane at N = 512: 4 x 183 + 64 + 183 = 980 VALU instructions for
 * 1792 input samples, against 1792 packed FMAs in the direct form.
 *
 * Layout: one wave per channel; lane t owns window quads t + 64 j (j < P), exactly the
 * x[t + j NT] the first FFT pass wants, so the input goes from the 16-byte global loads straight
 * into the transform's registers -- no polyphase planes in LDS.  Consecutive windows overlap by 64
 * quads (the 256 raw samples of the FIR history): the j = P-1 quads of one frame are the j = 0
 * quads of the next and stay in registers; HBM is still read exactly once.  Decimated samples go
 * into a ring in LDS from which the overlap-save frames (front_frame) take N/2 at a time.
 *
 * Two frame lengths (template VC, new quad columns per frame; state is the same 256 raw samples as the
 * direct form in both):
 *   VC = 7 (fir_variant 2, bench.py): 448 outputs per 512-point window.  Frames are anchored at the call's first
 *     sample and the last one of a call is partial (inputs past the end of the call are zeros; every output depends
 *     on inputs at or before its own time only, so the valid ones are exact).  A stream cut into calls differently
 *     rounds differently (the frame grid moves).
 *   VC = 4 (the library's default): ONE GRANULE per frame -- 256 outputs, the window's last three columns zeros.
 *     Every call boundary is a frame boundary and a frame's input is a function of the absolute sample position:
 *     the same bits for any call split, at 5 transforms per 256 outputs instead of per 448.
 * Pipelining, sub-batches and the channel partition never change a bit in either.  The pre-processor's IQ swap and
 * the noise blanker are compiled in with PRE. */
template <int N, int P, bool LEAN, bool PRE, bool Q4 = false, int VC = RDSP_FD_P - 1>
__global__ void __launch_bounds__(N / P, 2) rdsp_front_fd_kernel(RdspFrontParams p) {
  using PL = FftPlan<N, P>;              /* the overlap-save filter's transform (FFT_L)     */
  constexpr int ND = RDSP_FD_N, PD = RDSP_FD_P; /* the decimator's: 512 points whatever FFT_L is  */
  using PLD = FftPlan<ND, PD>;
  constexpr int NT = PL::NT;
  constexpr int H = N / 2;
  constexpr int PH = P / 2;
  /* VC: quad columns of new input per decimator frame.  7 (448 outputs of the 512-point window: the throughput
   * form, fir_variant 2) or 4: frames of ONE GRANULE -- 256 outputs, the window's last three columns zeros --
   * so that every call boundary is a frame boundary and every frame's input is a function of the absolute
   * sample position: the same bits for any call split (fir_variant 4, the library's default), at 5 transforms
   * per 256 outputs instead of per 448 */
  constexpr int VAL = 64 * VC; /* valid outputs per decimator frame */
  constexpr int NC = VC + 1;   /* data columns of a frame's window: the shared / history column and VC new ones */
  static_assert(VC == PD - 1 || VC == 4, "448-sample frames or one granule per frame");
  /* FFT_L >= 2048 runs four waves per channel: every wave takes its own decimator frame (four
   * frames per round, no sums across waves), then all of them share the overlap-save frames */
  constexpr int NW = NT / 64;
  /* FFT_L 256: four overlap-save frames per pass, a 16-lane row each (front_frame_quad) */
  constexpr bool QUAD = Q4;
  static_assert(!Q4 || N == 256, "the four-frame form exists for FFT_L 256");
  constexpr int RING = QUAD ? QUAD_RING : ((NW == 1) ? 1024 : 4096); /* >= (H - 1) + NW * VAL, power of two (QUAD: eight padded hops) */
  constexpr bool SAME = (N == ND && P == PD);   /* one plan: twiddles and LDS bases are shared */
  static_assert((NT == 64 || NT == 256) && PLD::NT == 64, "one or four waves per channel, one per decimator frame");
  static_assert(QUAD || H - 1 + NW * VAL <= RING, "ring holds a round's outputs behind an unfinished hop");
  static_assert(!QUAD || H + (4 * H - 64) + VAL <= QUAD_HOPS * H, "the overlap hop and what is unconsumed (< 4 hops, in steps of 64) survive a frame's seven columns");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *ring = reinterpret_cast<float2 *>(smem_raw);
  float2 *wb = ring + RING;
  constexpr int WBN0 = PL::WB > NW * PLD::WB ? PL::WB : NW * PLD::WB;
  constexpr int WBN = (QUAD && QUAD_WB > WBN0) ? QUAD_WB : WBN0;
  float *red = reinterpret_cast<float *>(wb + WBN);

  const bool SWAP_IQ = PRE && p.swap_iq != 0;
  /* the noise blanker takes the quad columns in stream order.  With four waves per channel the
   * frames of a round run side by side, so the blanker's pre-pass goes round the waves in frame
   * order before the transforms start: its state (level, per-lane window sums) and every frame's
   * last column as blanked (the next frame's column 0) are handed on through LDS */
  const bool NB_ON = PRE && p.nb_on != 0;
  uint4 *nbcol = reinterpret_cast<uint4 *>(red + 64);   /* [NW][64] (four-wave kernels only) */
  float *nbacc = reinterpret_cast<float *>(nbcol + NW * 64); /* [64] per-lane sums of the open window */
  float *nbs = nbacc + 64;                              /* [0]: level */
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  float2 *wbd = wb + wave * PLD::WB; /* this wave's decimator work buffer (inside the filter's) */
  const size_t ch = (size_t)p.ch_base + blockIdx.x;
  const uint32_t *iq = p.iq + ch * p.in_stride;
  RdspGroup G; /* its hot fields; what only the call's first frame needs is read there (RDSP_GROUP_LATE_*) */
  const uint32_t gi = p.group_of ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.group_of[ch]) : 0u;
  {
    const uint32_t *gw = reinterpret_cast<const uint32_t *>(p.groups + gi);
    uint32_t r[32];
#pragma unroll
    for (int i = 0; i < 32; i++) r[i] = (i < 30) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)gw[i]) : 0u;
    G = __builtin_bit_cast(RdspGroup, r);
  }
  const int total = p.n_chunks * 256; /* outputs = input quads of this call */

  /* raw quads of this wave's first frame (frame `wave`): column j holds quads
   * fr*VAL - 64 + lane + 64 j; for frame 0 column 0 is the FIR history (the 64 quads before the call) */
  /* The call's input of this channel as a raw buffer: a quad past the end of the call reads as zeros by the
   * buffer's range check -- no compare, no exec-mask branch and no zeroed registers per load, and the
   * loads are unconditional, so the waits for the table loads issued before them are counted exactly
   * (behind a conditional load the compiler has to assume it was not issued, and every wait for an
   * older load became a wait for the whole prefetch: an HBM round trip inside the frame) */
  const __amdgpu_buffer_rsrc_t iq_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(iq), 0, 16 * total, 0x00020000);
  auto ld_quad = [&](int q) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    /* aux 2 = nt: the stream passes once (one wave per channel).  Four waves per channel re-read each other's
     * frame overlap, which they should find in L2: default policy there */
    const v4i v = __builtin_amdgcn_raw_buffer_load_b128(iq_rsrc, 16 * q, 0, NW == 1 ? 2 : 0);
    return make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
  };
  uint4 rq[NC];
#pragma unroll
  for (int j = 0; j < NC; j++) {
    const int q = wave * VAL - 64 + lane + 64 * j;
    if (q < 0) rq[j] = *reinterpret_cast<const uint4 *>(p.st_hist + ch * 256 + 4 * lane);
    else rq[j] = ld_quad(q);
  }

  Twiddles<N, P, LEAN> tw;
  LdsBases<N, P, false> lb;
  if constexpr (!QUAD) {
    tw.init(tid);
    make_lds_bases<N, P, false>(tid, lb);
  }
  /* FFT_L 256: the 16-point-per-lane plan of front_frame_quad, a lane's place in its row */
  Twiddles<256, 16, LEAN> tw16;
  LdsBases<256, 16, false> lb16;
  if constexpr (QUAD) {
    tw16.init(lane & 15);
    make_lds_bases<256, 16, false>(lane & 15, lb16);
  }
  const int mbase = 16 * (lane & 3) + 4 * ((lane >> 2) & 3); /* this lane's bins in the radix-4 plan's mask image */
  /* the decimator's plan: its own twiddles and LDS bases unless it is the filter's plan */
  Twiddles<ND, PD, false> twd_own;
  LdsBases<ND, PD, false> lbd_own;
  if constexpr (!SAME) {
    twd_own.init(lane);
    make_lds_bases<ND, PD, false>(lane, lbd_own);
  }
  const auto &twd = [&]() -> const auto & { if constexpr (SAME) return tw; else return twd_own; }();
  const auto &lbd = [&]() -> const auto & { if constexpr (SAME) return lb; else return lbd_own; }();
  uint32_t vadbits = 0;
  if constexpr (QUAD) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int k = (lane & 15) + 16 * e; /* bin_of_pos<256, 16>(16 i + e) */
      if (k >= p.vad_lo && k <= p.vad_hi) vadbits |= 1u << e;
    }
  } else {
#pragma unroll
    for (int e = 0; e < P; e++) {
      int k = bin_of_pos<N, P>(tid * P + e);
      if (k >= p.vad_lo && k <= p.vad_hi) vadbits |= 1u << e;
    }
  }
  float nfloor = p.st_scal[ch * 4 + 0];
  const float vad_inv = 1.0f / (float)(p.vad_hi - p.vad_lo);
  float agc_g = p.st_scal[ch * 4 + 1];
  float am_dc = p.st_scal[ch * 4 + 2];
  float nb_level = p.st_scal[ch * 4 + 3], nb_acc = 0.f;
  uint4 hist_save = make_uint4(0u, 0u, 0u, 0u); /* the call's last 64 quads as they entered the decimator */
  float2 vprev[PH];
  if constexpr (QUAD) { /* the previous hop goes in front of the ring's first one (the last of the ring) */
#pragma unroll
    for (int j = 0; j < PH; j++) ring[(QUAD_HOPS - 1) * QUAD_PITCH + tid + j * NT] = p.st_prev[ch * H + tid + j * NT];
  } else {
#pragma unroll
    for (int j = 0; j < PH; j++) vprev[j] = p.st_prev[ch * H + tid + j * NT];
  }
  int frame_idx = 0;
  int produced = 0, consumed = 0;
  int rhop = 0; /* QUAD: ring hop of the oldest unconsumed sample */
  auto wsync = []() { wg_sync<1>(); };
  if constexpr (NW > 1) {
    if (NB_ON) {
      if (tid < 64) nbacc[tid] = 0.f;
      if (tid == 0) nbs[0] = nb_level;
      wg_sync<NW>();
    }
  }

#pragma unroll 1
  for (int round = 0; produced < total; round++) {
    const int fr = round * NW + wave; /* this wave's frame; past the end of the call it works on zeros */
    if (NB_ON) {
      /* noise blanker (engine feature, build-defined): decision windows of 1024 input samples =
       * four quad columns; a frame brings seven new columns (j = 1..7), taken in stream order.  A
       * sample whose power exceeds the reference level x threshold is zeroed in the raw word, so it
       * stays blanked in the next frame's column 0 and in the FIR history; the level moves at the
       * end of every window from the mean post-blanking power (one wave reduction) */
      auto blank_frame = [&]() {
#pragma unroll
        for (int j = 1; j < NC; j++) {
          const int c = VC * fr + (j - 1); /* column of the call */
          if (64 * c < total) {
            const float thr = nb_level * p.nb_thr;
            uint32_t w[4] = {rq[j].x, rq[j].y, rq[j].z, rq[j].w};
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const uint32_t ww = SWAP_IQ ? __builtin_amdgcn_alignbit(w[r], w[r], 16) : w[r];
              const float2 x = unpack_iq(ww, p.scale_i, p.scale_q);
              const float pw = x.x * x.x + x.y * x.y;
              const bool blanked = nb_level > 0.f && pw > thr;
              w[r] = blanked ? 0u : w[r];
              nb_acc += blanked ? 0.f : pw;
            }
            rq[j] = make_uint4(w[0], w[1], w[2], w[3]);
            if ((c & 3) == 3) {
              const float mean = wave_sum(nb_acc) / 1024.0f;
              nb_level = (nb_level > 0.f) ? nb_level + 0.2f * (mean - nb_level) : mean;
              nb_acc = 0.f;
            }
            if (64 * (c + 1) == total) {
              if constexpr (NW == 1) hist_save = rq[j];
              else *reinterpret_cast<uint4 *>(RDSP_LATE(st_hist) + ch * 256 + 4 * lane) = rq[j]; /* the call's last 64 quads */
            }
          }
        }
      };
      if constexpr (NW == 1) {
        blank_frame();
      } else {
#pragma unroll 1
        for (int w = 0; w < NW; w++) {
          if (wave == w) {
            nb_level = nbs[0];
            nb_acc = nbacc[lane];
            if (fr > 0) rq[0] = nbcol[(w + NW - 1) % NW * 64 + lane]; /* the frame before, as blanked */
            blank_frame();
            nbcol[w * 64 + lane] = rq[VC];
            nbacc[lane] = nb_acc;
            if (lane == 0) nbs[0] = nb_level;
          }
          wg_sync<NW>();
        }
      }
    }
    /* ---- A2: phasors of this lane's P quad columns (sample 4 q + r of a quad follows by rot_r) */
    const uint32_t nq = p.n0 + 4u * (uint32_t)(fr * VAL - 64 + lane); /* absolute index of column 0 */
    const bool hist = (fr == 0);
    /* column 0 of the call's first frame is the previous call's samples: they keep the swap flag and
     * the gains they came in with (uniform values, chosen once per frame).  Only the PRE kernels carry
     * this: the launch code picks them for the one call after such a setting changed */
    float si0 = p.scale_i, sq0 = p.scale_q;
    bool swap0 = PRE && p.swap_iq != 0;
    uint32_t dphi_hist = G.dphi;
    if (hist) { /* round 0 only: read here, not held in scalar registers for the whole launch */
      if constexpr (PRE) {
        si0 = RDSP_LATE(scale_i_hist);
        sq0 = RDSP_LATE(scale_q_hist);
        swap0 = RDSP_LATE(swap_hist) != 0;
      }
      dphi_hist = RDSP_GROUP_LATE_U32(gi, dphi_hist);
    }
    /* Gains.  A column whose I and Q gains are equal carries its gain on the phasor (two packed multiplies per
     * frame instead of 32 on the samples; x (g ph) = (x g) ph to the bit when g is a power of two -- unit input
     * gain -- and to an ulp otherwise); a column with two gains (IQ balance) is scaled per sample.  The rule
     * looks at the column's own gains only, in the kernels with and without PRE alike (without PRE the launch
     * code guarantees one gain, history included), so how a sample rounds does not depend on which of the two
     * kernels a call split happens to run it through */
    const bool fold = !PRE || p.scale_i == p.scale_q, fold0 = !PRE || si0 == sq0;
    const float gph = fold ? p.scale_i : 1.0f, gph0 = fold0 ? si0 : 1.0f;      /* on the phasor ...           */
    const float sxi = fold ? 1.0f : p.scale_i, sxq = fold ? 1.0f : p.scale_q;  /* ... or on the samples (x 1.0 is exact) */
    const float sxi0 = fold0 ? 1.0f : si0, sxq0 = fold0 ? 1.0f : sq0;
    float2 pj[NC];
    {
      float2 b1u = make_float2(1.f, 0.f);
      if (G.dphi != 0u) b1u = nco_phasor_alu((nq + 256u) * G.dphi);
      const float2 b1 = make_float2(b1u.x * gph, b1u.y * gph); /* the gain first, the rotations after it */
      /* column 0: one column before b1, evaluated the same way in every frame -- a frame's phasors are a
       * function of its absolute position (and the column's own gain), not of where the call began.  Only
       * behind a retune (the previous call's samples were mixed with another increment) it is evaluated
       * directly with that increment */
      if (hist && dphi_hist != G.dphi) {
        const float2 d = (dphi_hist != 0u) ? nco_phasor_alu(nq * dphi_hist) : make_float2(1.f, 0.f);
        pj[0] = make_float2(d.x * gph0, d.y * gph0);
      } else if (PRE && gph0 != gph) {
        pj[0] = cmulc_uniform(make_float2(b1u.x * gph0, b1u.y * gph0), G.rotq1);
      } else {
        pj[0] = cmulc_uniform(b1, G.rotq1);
      }
      pj[1] = b1;
      if constexpr (PD > 2) pj[2] = cmul_pinned_u(b1, G.rotq1);
      if constexpr (PD > 3) pj[3] = cmul_pinned_u(b1, G.rotq2);
#pragma unroll
      for (int j = 4; j < NC; j++) pj[j] = cmul_pinned_u(pj[j - 3], G.rotq3);
    }

    /* ---- A1 + A3: four branch transforms, multiply-accumulate with the branch spectra ------- */
    float2 acc[PD];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float2 gm[PD]; /* G_r slice of this lane: L2-resident, lands behind the transform */
      {
        const float2 *mp = p.fd_mask + (size_t)r * ND;
        asm volatile("" : "+s"(mp));
        const auto gp = as_global(mp);
#pragma unroll
        for (int e = 0; e < PD; e++) gm[e] = gp[e * 64 + lane];
      }
      float2 v[PD];
#pragma unroll
      for (int j = NC; j < PD; j++) v[j] = make_float2(0.f, 0.f); /* granule frames: the rest of the window is zeros */
#pragma unroll
      for (int j = 0; j < NC; j++) {
        uint32_t w = (r == 0) ? rq[j].x : (r == 1) ? rq[j].y : (r == 2) ? rq[j].z : rq[j].w;
        if (j == 0 ? swap0 : SWAP_IQ) w = __builtin_amdgcn_alignbit(w, w, 16);
        float2 x = make_float2((float)(int16_t)(w & 0xFFFFu), (float)(int16_t)(w >> 16));
        if constexpr (PRE) x = make_float2(x.x * (j == 0 ? sxi0 : sxi), x.y * (j == 0 ? sxq0 : sxq));
        float2 ph = pj[j];
        if (r > 0) {
          const float2 rr = (r == 1) ? G.rot1 : (r == 2) ? G.rot2 : G.rot3;
          if (j == 0) { /* the history column of the call's first frame: the rotation it was mixed with */
            float2 r0 = rr;
            if (hist) r0 = (r == 1) ? RDSP_GROUP_LATE_F2(gi, roth1) : (r == 2) ? RDSP_GROUP_LATE_F2(gi, roth2) : RDSP_GROUP_LATE_F2(gi, roth3);
            ph = cmul_pinned_u(ph, r0);
          } else {
            ph = cmul_pinned_u(ph, rr);
          }
        }
        v[j] = cmul_pinned(x, ph);
      }
      if (r == 3) { /* the raw registers are free: the next frame's loads land behind the transforms */
        if constexpr (NW == 1) rq[0] = rq[VC]; /* consecutive frames share a column */
#pragma unroll
        for (int j = (NW == 1 ? 1 : 0); j < NC; j++) {
          const int q = (fr + NW) * VAL - 64 + lane + 64 * j; /* >= 0: this is frame 1 or later */
          rq[j] = ld_quad(q);
        }
      }
      {
        float2 twp[PD - 1];
        twd.template get<0>(twp);
        fwd_pass0_store<ND, PD>(lbd, v, wbd, twp);
      }
      /* the decimator's transforms run in a work buffer of the wave's own (wbd), also with four waves per
       * channel: the lanes of ONE wave are all that has to be ordered here.  Between waves the barriers are the
       * one behind the ring writes below and the one that ends every overlap-save frame ("wb is free again") */
      wg_sync<1>();
      fwd_mid_all<ND, PD, 1, PLD::NP - 1, false>(lbd, wbd, twd, wsync);
      fwd_pass_last<ND, PD>(lbd, v, wbd);
      wg_sync<1>(); /* wbd is rewritten by the next branch */
#pragma unroll
      for (int e = 0; e < PD; e++) acc[e] = (r == 0) ? cmul(v[e], gm[e]) : cmac(acc[e], v[e], gm[e]);
    }
    inv_pass_last<ND, PD>(lbd, acc, wbd);
    wg_sync<1>();
    inv_mid_all<ND, PD, PLD::NP - 2, false>(lbd, wbd, twd, wsync);
    {
      float2 twp[PD - 1];
      twd.template get<0>(twp);
      inv_pass0_load<ND, PD>(lbd, acc, wbd, twp);
    }
    /* acc[j] = y at window index lane + 64 j; index 64 (j = 1) is output fr*VAL of the call */
    {
      /* a frame's outputs start at a multiple of 64 in the ring, so a column of 64 never wraps: the wrap is
       * scalar arithmetic per column, one vector add per store (past the end of the call: slots nobody
       * consumes, `produced` stops at total) */
      if constexpr (QUAD) { /* columns of 64 in padded hops */
        const int w0 = __builtin_amdgcn_readfirstlane((fr * VC) % (2 * QUAD_HOPS));
#pragma unroll
        for (int j = 1; j < NC; j++) {
          int cw = w0 + (j - 1);
          cw = cw >= 2 * QUAD_HOPS ? cw - 2 * QUAD_HOPS : cw;
          ring[cw * 64 + (cw >> 1) * (QUAD_PITCH - 128) + lane] = acc[j];
        }
      } else {
        const int m0 = __builtin_amdgcn_readfirstlane(fr * VAL);
        static_assert(VAL % 64 == 0 && (QUAD || RING % 64 == 0), "ring columns");
#pragma unroll
        for (int j = 1; j < NC; j++) ring[((m0 + 64 * (j - 1)) & (RING - 1)) + lane] = acc[j];
      }
    }
    produced = (round + 1) * NW * VAL < total ? (round + 1) * NW * VAL : total;
    wg_sync<NW>();

    /* ---- A5/A6: overlap-save frames over what the ring holds ------------------------------ */
    if constexpr (QUAD) {
      /* four at a time; at the end of the call whatever is left (frames are complete hops: the call is whole granules) */
#pragma unroll 1
      while (produced - consumed >= 4 * H || (produced == total && produced - consumed >= H)) {
        int nf = (produced - consumed) / H;
        nf = nf > 4 ? 4 : nf;
        front_frame_quad(p, G, tw16, lb16, wb + (lane >> 4) * FftPlan<256, 16>::WB, ring, rhop, nf, mbase, vadbits, vad_inv,
                         nfloor, agc_g, am_dc, frame_idx, ch, lane);
        frame_idx += nf;
        consumed += nf * H;
        rhop += nf;
        rhop = rhop >= QUAD_HOPS ? rhop - QUAD_HOPS : rhop;
      }
      continue;
    }
#pragma unroll 1
    while (produced - consumed >= H) {
      float2 mreg[P];
      {
        const float2 *mp = p.mask_pool + G.mask_off;
        asm volatile("" : "+s"(mp));
        const auto gp = as_global(mp);
#pragma unroll
        for (int e = 0; e < P; e++) mreg[e] = gp[e * NT + tid];
      }
      /* hops start at multiples of H in a ring of a multiple of H: the hop is contiguous */
      static_assert(QUAD || RING % H == 0, "a hop never wraps");
      const float2 *hop = ring + (consumed & (RING - 1));
      front_frame<N, P, false>(p, G, tw, lb, wb, red, mreg, vadbits, vad_inv, vprev, nfloor, agc_g, am_dc, frame_idx, ch,
                               tid, [&](int i) { return hop[i]; });
      consumed += H;
    }
  }

  /* ---- state out: previous hop, the last 256 raw samples (an L2 re-read), scalars --------- */
  float2 *const st_prev = RDSP_LATE(st_prev); /* the state pointers again: not kept across the frame loop */
  uint32_t *const st_hist = RDSP_LATE(st_hist);
  float *const st_scal = RDSP_LATE(st_scal);
  if constexpr (QUAD) {
    const int hp = rhop == 0 ? QUAD_HOPS - 1 : rhop - 1; /* the last hop consumed */
#pragma unroll
    for (int j = 0; j < PH; j++) st_prev[ch * H + tid + j * NT] = ring[hp * QUAD_PITCH + tid + j * NT];
  } else {
#pragma unroll
    for (int j = 0; j < PH; j++) st_prev[ch * H + tid + j * NT] = vprev[j];
  }
  if (tid < 64 && !(NB_ON && NW > 1)) /* four waves with the blanker: stored by the wave that blanked them */
    *reinterpret_cast<uint4 *>(st_hist + ch * 256 + 4 * tid) =
        NB_ON ? hist_save : *reinterpret_cast<const uint4 *>(iq + 4 * (total - 64 + tid));
  if (tid == 0) {
    st_scal[ch * 4 + 0] = nfloor;
    if (!p.to_mid) st_scal[ch * 4 + 1] = agc_g;
    st_scal[ch * 4 + 2] = am_dc;
    if (NB_ON) st_scal[ch * 4 + 3] = (NW > 1) ? nbs[0] : nb_level;
  }
}

template <int N, int P, bool Q4 = false>
constexpr size_t front_fd_lds() {
  constexpr int nw = N / P / 64;
  constexpr int wbd = nw * FftPlan<RDSP_FD_N, RDSP_FD_P>::WB;
  constexpr int wbn0 = FftPlan<N, P>::WB > wbd ? FftPlan<N, P>::WB : wbd;
  constexpr int wbn = (Q4 && QUAD_WB > wbn0) ? QUAD_WB : wbn0;
  /* + the blanker's hand-over area of the four-wave kernels: [nw][64] quads, 64 sums, the level */
  return (size_t)((Q4 ? QUAD_RING : (nw == 1 ? 1024 : 4096)) + wbn) * sizeof(float2) + 64 * sizeof(float) +
         (nw > 1 ? (size_t)nw * 64 * sizeof(uint4) + 64 * sizeof(float) + 16 : 0);
}

/* ---- front kernel with the frequency-domain decimator on 16-lane rows (round 6) -------------
 * The same polyphase overlap-save decimator as rdsp_front_fd_kernel -- four low-rate forward transforms of the
 * mixed input, branch spectra, one inverse -- on 256-point windows, one window per 16-lane DPP row: 16 points per
 * lane, two radix-16 passes, ONE LDS exchange each way (the plan of front_frame_quad), a wave taking four
 * consecutive windows at once.  A window is the 64 quads in front of its frame (the 256 raw samples of the FIR
 * history; rows and passes re-read them, an L2 hit) and RV new ones:
 *   RV = 128 (fir_variant 5): two frames per granule of 256 outputs, the window's last quarter zeros.  Every call
 *     boundary is a frame boundary and a frame's arithmetic does not depend on the row or pass it lands in: the
 *     same bits for any call split, like the one-granule form of rdsp_front_fd_kernel;
 *   RV = 192 (EXPERIMENTAL=1 builds, fir_variant 6): the whole window is data; frames anchored at the call's first
 *     sample, the last one partial.
 * Mixer: one phasor per lane and pass (its first new column), every (column, branch) by one product with an entry
 * of a 64-entry table in LDS, exp(-j theta (64 (j - 4) + r)), made at the start of the launch.
 * What it costs, from the ISA: a row's transform is 184 packed instructions for 16 points (2 x 77 + 15 twiddle
 * products), the wave-wide 512-point radix-8 one 113 for 8: 19 % less per point, which the shorter window gives
 * back -- 1330 packed instructions per pass of 768 (512) outputs against 759 per frame of 448 (256).  Measured at
 * K2 / K4 (PMC and same-box A/B, tests/micro/rows_ab.sh, rows_pmc.sh): RV 128 0.727 / 2.04 ms per step where the
 * one-granule form takes 0.808 / 2.10 and 448-sample frames 0.598 / 1.74; RV 192 0.742 / 2.05 (as many vector
 * instructions as the 448-sample form, 2.19e8 against 2.16e8 per K2 launch, and 38 spilled registers).  So: an
 * opt-in for chains that want split-invariant bits and whose audio does not go on to a tail kernel on the same
 * SIMDs (238-256 VGPRs where the wave-wide forms fit 176; K3 pipelined: 1.39-1.43 ms against 1.38-1.46).
 * The noise blanker (whose decisions go with the raw words from frame to frame) runs in rdsp_front_fd_kernel:
 * the launch code falls back to the form with the same split behaviour. */
constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr int RD_WB = 4 * FftPlan<256, 16>::WB; /* a wave's four row exchange buffers */

template <int N, int P, bool LEAN, bool PRE, bool Q4, int RV>
__global__ void __launch_bounds__(N / P, 2) rdsp_front_rd_kernel(RdspFrontParams p) {
  using PL = FftPlan<N, P>;
  using PR = FftPlan<256, 16>;
  constexpr int NT = PL::NT;
  constexpr int NW = NT / 64;
  constexpr int H = N / 2;
  constexpr int PH = P / 2;
  static_assert(RV == 128 || RV == 192, "two frames per granule, or the whole window");
  constexpr int NJ = 4 + RV / 16; /* data points of a lane: four history columns and RV / 16 new ones */
  constexpr int PASS = 4 * RV;    /* outputs of one wave's pass */
  constexpr bool QUAD = Q4;
  static_assert(!Q4 || N == 256, "the four-frame form exists for FFT_L 256");
  /* QUAD ring: the overlap hop, what a pass leaves unconsumed (< 4 hops, whole hops: 0 or 2 at RV 192) and a pass */
  constexpr int HOPS = RV == 128 ? QUAD_HOPS : 9;
  static_assert(!QUAD || 1 + (RV == 128 ? 0 : 2) + PASS / 128 <= HOPS, "ring hops");
  constexpr int RING = QUAD ? HOPS * QUAD_PITCH : ((NW == 1) ? 1024 : 4096);
  /* what a round leaves unconsumed is a multiple of gcd(H, NW PASS) below H */
  static_assert(QUAD || (H - cgcd(H, NW * PASS)) + NW * PASS <= RING, "ring holds a round's outputs behind an unfinished hop");
  static_assert(NT == 64 || NT == 256, "one or four waves per channel, a pass of four frames each");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2 *ring = reinterpret_cast<float2 *>(smem_raw);
  float2 *wb = ring + RING;
  constexpr int WBN0 = PL::WB > NW * RD_WB ? PL::WB : NW * RD_WB;
  constexpr int WBN = (QUAD && QUAD_WB > WBN0) ? QUAD_WB : WBN0;
  float *red = reinterpret_cast<float *>(wb + WBN);
  float2 *utab = reinterpret_cast<float2 *>(red + 64); /* [16][4] mixer rotations by 64 (j - 4) + r samples */

  const bool SWAP_IQ = PRE && p.swap_iq != 0;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int row = lane >> 4, li = lane & 15;
  float2 *wbg = wb + wave * RD_WB + row * PR::WB; /* this row's exchange buffer (inside the filter's work buffer) */
  const size_t ch = (size_t)p.ch_base + blockIdx.x;
  const uint32_t *iq = p.iq + ch * p.in_stride;
  RdspGroup G; /* its hot fields */
  const uint32_t gi = p.group_of ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.group_of[ch]) : 0u;
  {
    const uint32_t *gw = reinterpret_cast<const uint32_t *>(p.groups + gi);
    uint32_t r[32];
#pragma unroll
    for (int i = 0; i < 32; i++) r[i] = (i < 30) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)gw[i]) : 0u;
    G = __builtin_bit_cast(RdspGroup, r);
  }
  const int total = p.n_chunks * 256; /* outputs = input quads of this call */
  if (tid < 64) {
    const int j = tid >> 2, r = tid & 3;
    utab[tid] = (G.dphi != 0u) ? nco_phasor_alu((uint32_t)(64 * (j - 4) + r) * G.dphi) : make_float2(1.f, 0.f);
  }

  const __amdgpu_buffer_rsrc_t iq_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(iq), 0, 16 * total, 0x00020000);
  auto ld_quad = [&](int q, bool once) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    /* aux 2 = nt for the columns nobody reads again; a frame's last 64 quads are the next one's history */
    const v4i v = once ? __builtin_amdgcn_raw_buffer_load_b128(iq_rsrc, 16 * q, 0, 2)
                       : __builtin_amdgcn_raw_buffer_load_b128(iq_rsrc, 16 * q, 0, 0);
    return make_uint4((uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w);
  };
  /* raw quads of this wave's first pass: frame 4 ps + row, window quads frame * RV - 64 + li + 16 j */
  uint4 rq[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) {
    const int q = (4 * wave + row) * RV - 64 + li + 16 * j;
    if (q < 0) rq[j] = *reinterpret_cast<const uint4 *>(p.st_hist + ch * 256 + 4 * (q + 64));
    else rq[j] = ld_quad(q, NW == 1 && j >= 4 && j < NJ - 4);
  }

  Twiddles<N, P, LEAN> tw;
  LdsBases<N, P, false> lb;
  if constexpr (!QUAD) {
    tw.init(tid);
    make_lds_bases<N, P, false>(tid, lb);
  }
  /* the rows' plan (and front_frame_quad's): kept in full where the filter stage uses it too */
  Twiddles<256, 16, QUAD ? LEAN : true> tw16;
  LdsBases<256, 16, false> lb16;
  tw16.init(li);
  make_lds_bases<256, 16, false>(li, lb16);
  const int mbase = 16 * (lane & 3) + 4 * ((lane >> 2) & 3);
  uint32_t vadbits = 0;
  if constexpr (QUAD) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int k = li + 16 * e;
      if (k >= p.vad_lo && k <= p.vad_hi) vadbits |= 1u << e;
    }
  } else {
#pragma unroll
    for (int e = 0; e < P; e++) {
      int k = bin_of_pos<N, P>(tid * P + e);
      if (k >= p.vad_lo && k <= p.vad_hi) vadbits |= 1u << e;
    }
  }
  float nfloor = p.st_scal[ch * 4 + 0];
  const float vad_inv = 1.0f / (float)(p.vad_hi - p.vad_lo);
  float agc_g = p.st_scal[ch * 4 + 1];
  float am_dc = p.st_scal[ch * 4 + 2];
  float2 vprev[PH];
  if constexpr (QUAD) {
#pragma unroll
    for (int j = 0; j < PH; j++) ring[(HOPS - 1) * QUAD_PITCH + tid + j * NT] = p.st_prev[ch * H + tid + j * NT];
  } else {
#pragma unroll
    for (int j = 0; j < PH; j++) vprev[j] = p.st_prev[ch * H + tid + j * NT];
  }
  int frame_idx = 0;
  int produced = 0, consumed = 0;
  int rhop = 0; /* QUAD: ring hop of the oldest unconsumed sample */
  int whop = 0; /* QUAD: ring hop the next pass's first output goes to */
  wg_sync<NW>(); /* utab */

  /* gains: as rdsp_front_fd_kernel -- one gain rides on the phasor, two (IQ balance) on the samples; the call's
   * first 64 quads (row 0 of pass 0, columns 0..3) keep the gains, the swap flag and the increment they came in with */
  const bool fold = !PRE || p.scale_i == p.scale_q;
  const float gph = fold ? p.scale_i : 1.0f;
  const float sxi = fold ? 1.0f : p.scale_i, sxq = fold ? 1.0f : p.scale_q;

#pragma unroll 1
  for (int round = 0; produced < total; round++) {
    const int ps = round * NW + wave; /* this wave's pass; past the end of the call it works on zeros */
    const int fr = 4 * ps + row;      /* this row's frame */
    const uint32_t nq = p.n0 + 4u * (uint32_t)(fr * RV + li); /* absolute index of column 4, branch 0 */
    float2 Bu = make_float2(1.f, 0.f);
    if (G.dphi != 0u) Bu = nco_phasor_alu(nq * G.dphi);
    const float2 B = make_float2(Bu.x * gph, Bu.y * gph); /* the gain first, the rotations after it */
    /* The call's first 64 quads (row 0 of pass 0, columns 0..3) came in under the previous call's settings.  They take
     * the same operations as every other column, with their own gain / swap flag, so that nothing rounds differently
     * when the settings did not change; only behind a retune their phasors are evaluated directly with the increment
     * they were mixed with.  PRE kernels and retunes only: the launch code picks them for the call after a change. */
    uint32_t dphi_hist = G.dphi;
    float si0 = p.scale_i, sq0 = p.scale_q;
    bool swap0 = SWAP_IQ;
    if (ps == 0) { /* read here (RDSP_LATE): not held in scalar registers for the whole launch */
      dphi_hist = RDSP_GROUP_LATE_U32(gi, dphi_hist);
      if constexpr (PRE) {
        si0 = RDSP_LATE(scale_i_hist);
        sq0 = RDSP_LATE(scale_q_hist);
        swap0 = RDSP_LATE(swap_hist) != 0;
      }
    }
    const bool fold0 = !PRE || si0 == sq0;
    const float gph0 = fold0 ? si0 : 1.0f, sxi0 = fold0 ? 1.0f : si0, sxq0 = fold0 ? 1.0f : sq0;
    const bool first_special = ps == 0 && (PRE || dphi_hist != G.dphi);
    const bool hl = ps == 0 && row == 0;
    const float gh = hl ? gph0 : gph;
    const float2 Bh = make_float2(Bu.x * gh, Bu.y * gh);

    float2 acc[16];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float2 gm[16]; /* G_r, bins li + 16 e: L2-resident, lands behind the transform */
      {
        const float2 *mp = p.rd_mask + (size_t)r * 256;
        asm volatile("" : "+s"(mp));
        const auto gp = as_global(mp);
#pragma unroll
        for (int e = 0; e < 16; e++) gm[e] = gp[e * 16 + li];
      }
      float2 v[16];
#pragma unroll
      for (int j = NJ; j < 16; j++) v[j] = make_float2(0.f, 0.f);
      auto word = [&](int j) { return (r == 0) ? rq[j].x : (r == 1) ? rq[j].y : (r == 2) ? rq[j].z : rq[j].w; };
      if (first_special) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint32_t w = word(j);
          if (hl ? swap0 : SWAP_IQ) w = __builtin_amdgcn_alignbit(w, w, 16);
          float2 x = make_float2((float)(int16_t)(w & 0xFFFFu), (float)(int16_t)(w >> 16));
          if constexpr (PRE) x = make_float2(x.x * (hl ? sxi0 : sxi), x.y * (hl ? sxq0 : sxq));
          float2 ph = cmul_pinned(Bh, lds_ld(&utab[4 * j + r]));
          if (dphi_hist != G.dphi) {
            float2 d = make_float2(1.f, 0.f);
            if (dphi_hist != 0u) d = nco_phasor_alu((nq + (uint32_t)(64 * (j - 4) + r)) * dphi_hist);
            ph = hl ? make_float2(d.x * gph0, d.y * gph0) : ph;
          }
          v[j] = cmul_pinned(x, ph);
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        if (j < 4 && first_special) continue;
        uint32_t w = word(j);
        if (SWAP_IQ) w = __builtin_amdgcn_alignbit(w, w, 16);
        float2 x = make_float2((float)(int16_t)(w & 0xFFFFu), (float)(int16_t)(w >> 16));
        if constexpr (PRE) x = make_float2(x.x * sxi, x.y * sxq);
        const float2 ph = cmul_pinned(B, lds_ld(&utab[4 * j + r]));
        v[j] = cmul_pinned(x, ph);
      }
      if (r == 3) { /* the raw registers are free: the next pass's loads land behind the transforms */
#pragma unroll
        for (int j = 0; j < NJ; j++) {
          const int q = (4 * (ps + NW) + row) * RV - 64 + li + 16 * j; /* >= 0: pass 1 or later */
          rq[j] = ld_quad(q, NW == 1 && j >= 4 && j < NJ - 4);
        }
      }
      {
        float2 twp[15];
        tw16.template get<0>(twp);
        fwd_pass0_store<256, 16>(lb16, v, wbg, twp);
      }
      wg_sync<1>(); /* a row's exchange buffer is its own */
      fwd_pass_last<256, 16>(lb16, v, wbg);
      wg_sync<1>();
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = (r == 0) ? cmul(v[e], gm[e]) : cmac(acc[e], v[e], gm[e]);
    }
    inv_pass_last<256, 16>(lb16, acc, wbg);
    wg_sync<1>();
    {
      float2 twp[15];
      tw16.template get<0>(twp);
      inv_pass0_load<256, 16>(lb16, acc, wbg, twp);
    }
    /* acc[j] = y at window index li + 16 j; index 64 (j = 4) is output fr * RV of the call */
    if constexpr (QUAD) {
      if constexpr (RV == 128) { /* a frame is a hop */
        int h = whop + wave * 4 + row;
        h = h >= HOPS ? h - HOPS : h;
        float2 *dst = ring + h * QUAD_PITCH + li;
#pragma unroll
        for (int c = 0; c < 8; c++) dst[16 * c] = acc[4 + c];
        whop = (whop + 4 * NW) % HOPS;
      } else { /* a frame is a hop and a half: even rows start a hop, odd ones in the middle of one */
        const int odd = row & 1;
        int a = whop + ((3 * row) >> 1);
        a = a >= HOPS ? a - HOPS : a;
        const int a1 = a + 1 >= HOPS ? a + 1 - HOPS : a + 1;
        float2 *dA = ring + a * QUAD_PITCH + (odd ? 64 : 0) + li;
        float2 *dB = odd ? ring + a1 * QUAD_PITCH + li : ring + a * QUAD_PITCH + 64 + li;
        float2 *dC = ring + a1 * QUAD_PITCH + (odd ? 64 : 0) + li;
#pragma unroll
        for (int c = 0; c < 4; c++) {
          dA[16 * c] = acc[4 + c];
          dB[16 * c] = acc[8 + c];
          dC[16 * c] = acc[12 + c];
        }
        whop = (whop + 6) % HOPS;
      }
    } else {
      const int mb = fr * RV; /* of the call; a multiple of 64 like the ring's length */
#pragma unroll
      for (int c = 0; c < RV / 16; c++) ring[((mb + 16 * c) & (RING - 1)) + li] = acc[4 + c];
    }
    produced = (round + 1) * NW * PASS < total ? (round + 1) * NW * PASS : total;
    wg_sync<NW>();

    /* ---- A5/A6: overlap-save frames over what the ring holds (as rdsp_front_fd_kernel) ------- */
    if constexpr (QUAD) {
#pragma unroll 1
      while (produced - consumed >= 4 * H || (produced == total && produced - consumed >= H)) {
        int nf = (produced - consumed) / H;
        nf = nf > 4 ? 4 : nf;
        front_frame_quad<HOPS>(p, G, tw16, lb16, wb + row * PR::WB, ring, rhop, nf, mbase, vadbits, vad_inv,
                               nfloor, agc_g, am_dc, frame_idx, ch, lane);
        frame_idx += nf;
        consumed += nf * H;
        rhop += nf;
        rhop = rhop >= HOPS ? rhop - HOPS : rhop;
      }
      continue;
    }
#pragma unroll 1
    while (produced - consumed >= H) {
      float2 mreg[P];
      {
        const float2 *mp = p.mask_pool + G.mask_off;
        asm volatile("" : "+s"(mp));
        const auto gp = as_global(mp);
#pragma unroll
        for (int e = 0; e < P; e++) mreg[e] = gp[e * NT + tid];
      }
      static_assert(QUAD || RING % H == 0, "a hop never wraps");
      const float2 *hop = ring + (consumed & (RING - 1));
      front_frame<N, P, false>(p, G, tw, lb, wb, red, mreg, vadbits, vad_inv, vprev, nfloor, agc_g, am_dc, frame_idx, ch,
                               tid, [&](int i) { return hop[i]; });
      consumed += H;
    }
  }

  /* ---- state out: previous hop, the last 256 raw samples (an L2 re-read), scalars --------- */
  float2 *const st_prev = RDSP_LATE(st_prev); /* the state pointers again: not kept across the frame loop */
  uint32_t *const st_hist = RDSP_LATE(st_hist);
  float *const st_scal = RDSP_LATE(st_scal);
  if constexpr (QUAD) {
    const int hp = rhop == 0 ? HOPS - 1 : rhop - 1; /* the last hop consumed */
#pragma unroll
    for (int j = 0; j < PH; j++) st_prev[ch * H + tid + j * NT] = ring[hp * QUAD_PITCH + tid + j * NT];
  } else {
#pragma unroll
    for (int j = 0; j < PH; j++) st_prev[ch * H + tid + j * NT] = vprev[j];
  }
  if (tid < 64)
    *reinterpret_cast<uint4 *>(st_hist + ch * 256 + 4 * tid) = *reinterpret_cast<const uint4 *>(iq + 4 * (total - 64 + tid));
  if (tid == 0) {
    st_scal[ch * 4 + 0] = nfloor;
    if (!p.to_mid) st_scal[ch * 4 + 1] = agc_g;
    st_scal[ch * 4 + 2] = am_dc;
  }
}

template <int N, int P, bool Q4, int RV>
constexpr size_t front_rd_lds() {
  constexpr int nw = N / P / 64;
  constexpr int wbn0 = FftPlan<N, P>::WB > nw * RD_WB ? FftPlan<N, P>::WB : nw * RD_WB;
  constexpr int wbn = (Q4 && QUAD_WB > wbn0) ? QUAD_WB : wbn0;
  constexpr int ringn = Q4 ? (RV == 128 ? QUAD_HOPS : 9) * QUAD_PITCH : (nw == 1 ? 1024 : 4096);
  return (size_t)(ringn + wbn + 64) * sizeof(float2) + 64 * sizeof(float);
}

/* one group record, rewritten in stream order (32 threads, one dword each) */
struct RdspGroupWords { uint32_t w[32]; };
__global__ void rdsp_group_store_kernel(uint32_t *dst, RdspGroupWords v) { dst[threadIdx.x] = v.w[threadIdx.x]; }

/* ---- pre-processor: I2S channel-slip correction (rdsp_pre_setIQslip) -------------------------
 * One rail of the stream is a sample behind the other: the corrected word pairs this sample's half
 * of one rail with the previous sample's half of the other (slip +1: I[n-1] | Q[n], slip -1:
 * I[n] | Q[n-1]).  A pass of its own in front of the front kernel, over the raw words: 8 bytes of HBM
 * traffic per input sample while the correction is on, nothing when it is off; the front kernels and
 * the 256-sample FIR history see corrected words only.  carry_in[ch * carry_stride]: the last raw word
 * of the previous call -- the chain's FIR history when that call ran without the correction, else the
 * word the previous pass left in carry_out (two arrays, alternating: every thread's predecessor word
 * is read before any carry is written). */
__global__ void __launch_bounds__(256) rdsp_iq_slip_kernel(const uint32_t *in, size_t in_stride, uint32_t *out,
                                                           size_t out_stride, const uint32_t *carry_in, size_t carry_stride,
                                                           uint32_t *carry_out, int n_quads, int slip) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  const size_t ch = blockIdx.y;
  if (q >= n_quads) return;
  const uint32_t *src = in + ch * in_stride;
  const uint4 w = *reinterpret_cast<const uint4 *>(src + 4 * (size_t)q);
  const uint32_t p = (q == 0) ? carry_in[ch * carry_stride] : src[4 * (size_t)q - 1];
  const uint32_t lo = 0x0000FFFFu;
  uint4 r;
  if (slip > 0) { /* I of the previous sample, Q of this one */
    r.x = (p & lo) | (w.x & ~lo); r.y = (w.x & lo) | (w.y & ~lo); r.z = (w.y & lo) | (w.z & ~lo); r.w = (w.z & lo) | (w.w & ~lo);
  } else {        /* I of this sample, Q of the previous one */
    r.x = (w.x & lo) | (p & ~lo); r.y = (w.y & lo) | (w.x & ~lo); r.z = (w.z & lo) | (w.y & ~lo); r.w = (w.w & lo) | (w.z & ~lo);
  }
  *reinterpret_cast<uint4 *>(out + ch * out_stride + 4 * (size_t)q) = r;
  if (q == n_quads - 1) carry_out[ch] = w.w;
}

/* ---- standalone A1 / A10 (bit-exact tests of the int16 <-> float edges) ---- */
__global__ void rdsp_q15_to_float_kernel(const int16_t *src, float *dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (float)src[i] * (1.0f / 32768.0f);
}
__global__ void rdsp_float_to_q15_kernel(const float *src, int16_t *dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (int16_t)q15_of_float(src[i]);
}

template <int N, int P, int DECIM, bool FMX>
constexpr size_t front_lds() {
  return FrontLds<N, P, DECIM, FMX>::BYTES;
}

template <int N, int P, int DECIM, bool LEAN, bool PRE, bool FMX>
int launch_front_x(const RdspFrontParams *p, int n_channels, hipStream_t stream) {
  constexpr size_t lds = front_lds<N, P, DECIM, FMX>();
  /* the raised dynamic-LDS limit is a per-device property of the function: one bit per device,
   * set under a lock (chains on several devices may launch from several host threads) */
  static std::mutex attr_mu;
  static uint64_t attr_done = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return (int)hipErrorInvalidDevice;
  {
    std::lock_guard<std::mutex> lk(attr_mu);
    if (!((attr_done >> dev) & 1u)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&rdsp_front_kernel<N, P, DECIM, LEAN, PRE, FMX>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
      attr_done |= (uint64_t)1 << dev;
    }
  }
  hipLaunchKernelGGL((rdsp_front_kernel<N, P, DECIM, LEAN, PRE, FMX>), dim3(n_channels), dim3(N / P), lds, stream, *p);
  return (int)hipGetLastError();
}
/* the raised dynamic-LDS limit is a per-device property of a kernel function: one bit per device,
 * set under a lock (chains on several devices may launch from several host threads) */
template <auto Kernel> /* one flag set per kernel instance */
int ensure_lds_limit(size_t lds) {
  static std::mutex mu;
  static uint64_t done = 0;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return (int)hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lk(mu);
  if (!((done >> dev) & 1u)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    done |= (uint64_t)1 << dev;
  }
  return 0;
}
/* Measurement switch RDSP_FD_LDS_PAD (bytes): unused LDS asked for on top of the one-wave frequency-domain kernels' own,
 * so that fewer of their workgroups fit on a compute unit beside the tail kernel (pipelined mode).  Both forms lose by
 * it (tests/micro/fd_lds_pad_sweep.sh, fd_lds_pad_sweep7.sh): the library never pads. */
inline size_t granule_form_lds_pad(int to_mid) {
  static const long env = getenv("RDSP_FD_LDS_PAD") ? atol(getenv("RDSP_FD_LDS_PAD")) : -1;
  if (env >= 0) return (size_t)env;
  (void)to_mid;
  return 0;
}
template <int N, int P, bool LEAN, bool PRE, int VC>
int launch_front_fd_vc(const RdspFrontParams *p, int n_channels, hipStream_t stream) {
  if constexpr (N == 256) {
    /* FFT_L 256: four overlap-save frames per pass (front_frame_quad) unless the audio goes on to the tail
     * kernel, which may share the SIMDs (pipelined mode) and leaves no room for that form's registers and LDS.
     * Measurement switch RDSP_NO_QUAD=1: the one-frame form here too (tests/micro/k2_occupancy.sh). */
    static const bool no_quad = getenv("RDSP_NO_QUAD") && atoi(getenv("RDSP_NO_QUAD")) != 0;
    if (!p->to_mid && !no_quad) {
      constexpr size_t lds4 = front_fd_lds<N, P, true>();
      static_assert(lds4 <= 48 * 1024, "no raised dynamic-LDS limit needed");
      hipLaunchKernelGGL((rdsp_front_fd_kernel<N, P, LEAN, PRE, true, VC>), dim3(n_channels), dim3(N / P), lds4, stream, *p);
      return (int)hipGetLastError();
    }
  }
  constexpr size_t lds = front_fd_lds<N, P>();
  if constexpr (lds > 48 * 1024) {
    int e = ensure_lds_limit<&rdsp_front_fd_kernel<N, P, LEAN, PRE, false, VC>>(lds);
    if (e != 0) return e;
  }
  size_t ask = lds;
  if constexpr (lds <= 16 * 1024) ask = lds + granule_form_lds_pad(p->to_mid);
  hipLaunchKernelGGL((rdsp_front_fd_kernel<N, P, LEAN, PRE, false, VC>), dim3(n_channels), dim3(N / P), ask, stream, *p);
  return (int)hipGetLastError();
}
/* the decimator on 16-lane rows (rdsp_front_rd_kernel); the plan's radix decides LEAN (the `lean` switch is for the
 * wave-wide forms) */
template <int N, int P, bool PRE, int RV>
int launch_front_rd(const RdspFrontParams *p, int n_channels, hipStream_t stream) {
  constexpr bool LEAN = (P >= 8); /* the filter's twiddles by product chains: the rows need the registers */
  if constexpr (N == 256) {
    if (!p->to_mid) { /* front_frame_quad behind it, as in launch_front_fd_vc */
      constexpr size_t lds4 = front_rd_lds<N, P, true, RV>();
      static_assert(lds4 <= 48 * 1024, "no raised dynamic-LDS limit needed");
      hipLaunchKernelGGL((rdsp_front_rd_kernel<N, P, LEAN, PRE, true, RV>), dim3(n_channels), dim3(N / P), lds4, stream, *p);
      return (int)hipGetLastError();
    }
  }
  constexpr size_t lds = front_rd_lds<N, P, false, RV>();
  if constexpr (lds > 48 * 1024) {
    int e = ensure_lds_limit<&rdsp_front_rd_kernel<N, P, LEAN, PRE, false, RV>>(lds);
    if (e != 0) return e;
  }
  hipLaunchKernelGGL((rdsp_front_rd_kernel<N, P, LEAN, PRE, false, RV>), dim3(n_channels), dim3(N / P), lds, stream, *p);
  return (int)hipGetLastError();
}
/* fir_fd 1: 448-sample frames (throughput form, fir_variant 2); 2: one granule per frame (split-invariant);
 * 3 / 4: the row forms with 128 / 192 outputs per window -- with the noise blanker on, the wave-wide form with the
 * same split behaviour (2 / 1): its decisions travel with the raw words from frame to frame */
template <int N, int P, bool LEAN, bool PRE>
int launch_front_fd(const RdspFrontParams *p, int n_channels, hipStream_t stream) {
  if (p->fir_fd >= 3 && !p->nb_on) {
    if (!p->rd_mask) return (int)hipErrorInvalidValue;
#ifdef RDSP_EXPERIMENTAL
    if (p->fir_fd == 4) return launch_front_rd<N, P, PRE, 192>(p, n_channels, stream);
#else
    if (p->fir_fd == 4) return (int)hipErrorNotSupported; /* 192 outputs per window: EXPERIMENTAL=1 builds (measured, no gain) */
#endif
    return launch_front_rd<N, P, PRE, 128>(p, n_channels, stream);
  }
  return (p->fir_fd == 2 || p->fir_fd == 3) ? launch_front_fd_vc<N, P, LEAN, PRE, 4>(p, n_channels, stream)
                                            : launch_front_fd_vc<N, P, LEAN, PRE, RDSP_FD_P - 1>(p, n_channels, stream);
}
template <int N, int P, int DECIM, bool LEAN, bool PRE>
int launch_front_w(const RdspFrontParams *p, int n_channels, hipStream_t stream) {
  if constexpr (DECIM == 4) {
    if (p->fir_fd) return launch_front_fd<N, P, LEAN, PRE>(p, n_channels, stream);
  }
#ifdef RDSP_EXPERIMENTAL
  if constexpr (DECIM == 4) {
    if (p->fir_matrix) return launch_front_x<N, P, DECIM, LEAN, PRE, true>(p, n_channels, stream);
  }
#else
  if (p->fir_matrix) return (int)hipErrorNotSupported; /* matrix-core FIR: EXPERIMENTAL=1 builds only */
#endif
  return launch_front_x<N, P, DECIM, LEAN, PRE, false>(p, n_channels, stream);
}
template <int N, int P, int DECIM, bool LEAN>
int launch_front_v(const RdspFrontParams *p, int n_channels, hipStream_t stream) {
  /* PRE: blanker, swap, or a FIR history that came in under another swap flag / other input gains */
  const bool hist_differs = p->swap_hist != p->swap_iq || p->scale_i_hist != p->scale_i || p->scale_q_hist != p->scale_q;
  /* ... or different gains on I and Q (iq_balance): the kernels without PRE fold the one gain into the mixer */
  return (p->nb_on || p->swap_iq || hist_differs || p->scale_i != p->scale_q) ? launch_front_w<N, P, DECIM, LEAN, true>(p, n_channels, stream)
                                  : launch_front_w<N, P, DECIM, LEAN, false>(p, n_channels, stream);
}
template <int N, int P, int DECIM>
int launch_front_t(const RdspFrontParams *p, int n_channels, hipStream_t stream) {
  if constexpr (P == 16) return launch_front_v<N, P, DECIM, true>(p, n_channels, stream);
  else return p->lean ? launch_front_v<N, P, DECIM, true>(p, n_channels, stream)
                      : launch_front_v<N, P, DECIM, false>(p, n_channels, stream);
}

}  // namespace

extern "C" size_t rdsp_front_lds_bytes(int fft_l, int decim) {
  const bool d4 = decim == 4;
  switch (fft_l) {
    case 256: return d4 ? front_lds<256, 4, 4, false>() : front_lds<256, 4, 1, false>();
    case 512: return d4 ? front_lds<512, 8, 4, false>() : front_lds<512, 8, 1, false>();
    case 1024: return d4 ? front_lds<1024, 16, 4, false>() : front_lds<1024, 16, 1, false>();
    case 2048: return d4 ? front_lds<2048, 8, 4, false>() : front_lds<2048, 8, 1, false>();
    case 4096: return d4 ? front_lds<4096, 16, 4, false>() : front_lds<4096, 16, 1, false>();
    default: return 0;
  }
}

extern "C" int rdsp_launch_front(int fft_l, int decim, const RdspFrontParams *p, int n_channels,
                                 hipStream_t stream) {
  if (decim != 1 && decim != 4) return (int)hipErrorInvalidValue;
  const bool d4 = decim == 4;
  switch (fft_l) {
    case 256: return d4 ? launch_front_t<256, 4, 4>(p, n_channels, stream) : launch_front_t<256, 4, 1>(p, n_channels, stream);
    case 512: return d4 ? launch_front_t<512, 8, 4>(p, n_channels, stream) : launch_front_t<512, 8, 1>(p, n_channels, stream);
    case 1024: return d4 ? launch_front_t<1024, 16, 4>(p, n_channels, stream) : launch_front_t<1024, 16, 1>(p, n_channels, stream);
    case 2048: return d4 ? launch_front_t<2048, 8, 4>(p, n_channels, stream) : launch_front_t<2048, 8, 1>(p, n_channels, stream);
    case 4096: return d4 ? launch_front_t<4096, 16, 4>(p, n_channels, stream) : launch_front_t<4096, 16, 1>(p, n_channels, stream);
    default: return (int)hipErrorInvalidValue;
  }
}

extern "C" int rdsp_launch_group_store(RdspGroup *dst, const RdspGroup *val, hipStream_t stream) {
  static_assert(sizeof(RdspGroup) == 128, "group record is 32 dwords");
  RdspGroupWords v;
  memcpy(&v, val, sizeof(v));
  hipLaunchKernelGGL(rdsp_group_store_kernel, dim3(1), dim3(32), 0, stream, reinterpret_cast<uint32_t *>(dst), v);
  return (int)hipGetLastError();
}

extern "C" int rdsp_launch_iq_slip(const uint32_t *in, size_t in_stride, uint32_t *out, size_t out_stride,
                                   const uint32_t *carry_in, size_t carry_stride, uint32_t *carry_out, int n_samples,
                                   int slip, int n_channels, hipStream_t stream) {
  const int nq = n_samples / 4;
  hipLaunchKernelGGL(rdsp_iq_slip_kernel, dim3((nq + 255) / 256, n_channels), dim3(256), 0, stream, in, in_stride, out,
                     out_stride, carry_in, carry_stride, carry_out, nq, slip);
  return (int)hipGetLastError();
}

extern "C" int rdsp_launch_q15_to_float(const int16_t *src, float *dst, size_t n, hipStream_t stream) {
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(rdsp_q15_to_float_kernel, dim3(grid), dim3(256), 0, stream, src, dst, n);
  return (int)hipGetLastError();
}
extern "C" int rdsp_launch_float_to_q15(const float *src, int16_t *dst, size_t n, hipStream_t stream) {
  int grid = (int)((n + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(rdsp_float_to_q15_kernel, dim3(grid), dim3(256), 0, stream, src, dst, n);
  return (int)hipGetLastError();
}
