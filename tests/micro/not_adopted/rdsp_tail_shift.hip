/*
 * rdsp_tail.hip -- the serial-in-time stages of the receive chain (gfx950):
 *
 *   rdsp_tail_shift_kernel<DUAL>      one channel per 16-lane DPP row, four per wave
 *       A7  NLMS noise reduction      RDSP_noise_reduction.h:35-80
 *       A8  ALS notch / peak          (AudioSDR, build-defined on A7's core)
 *       A9  AGC, output gain, A10 pack
 *
 * The NLMS recursion is serial in time: every step is a chain  dot product ->
 * 16-lane reduction -> error -> step size -> update, one wave per SIMD at 4096
 * channels.  Measured on MI355X (tests/micro/dpp_kinds.hip, tail_bench.hip): a lone wave
 * issues one instruction per ~5 cycles whatever its kind (a DPP add 4.4, a plain fp32 op 3.0,
 * a packed one 4.2 with three waves on the SIMD), so a step costs its instruction count and
 * its dependency chain.  The kernel uses a
 * one-step lookahead of the recursion,
 *        y_n = W_{n-1}.X_n + g_{n-1} (X_{n-1}.X_n) = A_n + g_{n-1} B_n,
 * which takes the reduction (A_n) off the g -> g chain (what stays loop-carried
 * is fma, sub, mul) and moves the energy E_n and the lag-1 correlation B_n, which
 * depend on the input only, into two DPP prefix scans per 16 steps.  The slot of
 * step n is written in a skewed order (finish the reduction of A_n, apply update
 * n-1, partial products of A_{n+1}, then y_n, e_n, g_n); pinning that order with
 * sched_barrier measured 5 % slower than letting the scheduler move within it.
 *
 * Compiled without the SLP vectoriser (see above: packing only adds moves here).
 */
#include "rdsp_wave.h"

using namespace rdsp;

namespace {

/* as dpp_f, lanes whose source falls outside the row read 0 (bound_ctrl) */
template <int CTRL>
__device__ __forceinline__ float dpp0_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
/* lane sub takes lane sub-1's value; lane 0 of the row keeps the DPP `old` operand = xin */
__device__ __forceinline__ float row_shift_in(float xin, float oldest) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, xin),
                                                               __builtin_bit_cast(int, oldest), 0x111, 0xF, 0xF, false));
}

#ifndef RDSP_TAIL_SWZ
#define RDSP_TAIL_SWZ 0 /* 1: the per-step reduction with ds_swizzle (LDS crossbar) instead of DPP */
#endif
/* butterfly partner inside the 16-lane row: DPP (fused into the add, 4.4 cycles of issue) or
 * ds_swizzle (LDS pipe, ~125 cycles of latency per stage; the add that follows is a plain VALU op) */
template <int STAGE>
__device__ __forceinline__ float red_partner(float v) {
#if RDSP_TAIL_SWZ
  constexpr int pat = ((1 << STAGE) << 10) | 0x1F; /* bit mode: lane ^ (1 << STAGE) */
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), pat));
#else
  return STAGE == 0 ? dpp_f<0xB1>(v) : STAGE == 1 ? dpp_f<0x4E>(v) : STAGE == 2 ? dpp_f<0x141>(v) : dpp_f<0x140>(v);
#endif
}

constexpr int LPC = 16;                  /* lanes per channel: one DPP row */
constexpr int TPL = RDSP_LMS_TAPS / LPC; /* taps per lane */
constexpr int NPH = 8;                   /* physical delay-line ring per lane (> TPL + 1, divides 128) */
constexpr int SCR = 48;                  /* per-group scalars: step size, B, energy */
static_assert(TPL == 6, "the slot schedule below is written for six taps per lane");

/* One NLMS instance of one channel.  Lane `sub` holds the taps of ages
 * TPL*sub .. TPL*sub+TPL-1 (age 0 = newest sample); CMSIS coefficient b[i]
 * multiplies age 95-i (arm_lms_norm_f32).  Measured alternatives on MI355X: 32 lanes
 * per channel (two waves per SIMD at 4096 channels) needs more instructions per
 * channel-step and ran 1.6x slower. */
struct Nlms {
  static constexpr int M = NPH - 1;
  float w[TPL];
  float xp[NPH];
  float energy;

  __device__ __forceinline__ void load(const float *wst, const float *prev, const float *est,
                                       size_t ch, int sub) {
#pragma unroll
    for (int k = 0; k < TPL; k++) w[k] = wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + k))];
#pragma unroll
    for (int k = 0; k < NPH; k++) xp[k] = 0.f;
    /* before step s the in-lane tap k sits at physical ((-s) + 1 + k) & M; a block
     * is 128 steps = a whole number of ring turns, so every block starts at s = 0 */
#pragma unroll
    for (int k = 0; k < TPL; k++) xp[(k + 1) & M] = prev[ch * RDSP_BLOCK + (127 - (TPL * sub + k))];
    energy = est[ch];
  }
  __device__ __forceinline__ void store(float *wst, float *est, size_t ch, int sub) {
#pragma unroll
    for (int k = 0; k < TPL; k++) wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + k))] = w[k];
    if (sub == 0) est[ch] = energy;
  }

  /* lane `sub` prepares step n = s0 + sub of a 16-step group: E_n, B_n by prefix
   * sums of their increments over the row, step size mu/(E_n + eps) */
  static __device__ __forceinline__ void prepare(const float *ring, int cb, int s0, int sub, float mu,
                                                 float e_base, float b_base, float *dst) {
    const int n = cb + s0 + sub;
    const float xn = ring[n & 255], xm = ring[(n - 1) & 255];
    const float xo = ring[(n - 96) & 255], xq = ring[(n - 97) & 255];
    float ea = fmaf(xn, xn, -(xo * xo)); /* E_n - E_{n-1}: arm_lms_norm_f32 energy update */
    float ba = fmaf(xn, xm, -(xo * xq)); /* B_n - B_{n-1} */
    ea += dpp0_f<0x111>(ea); ba += dpp0_f<0x111>(ba); /* row_shr 1, 2, 4, 8 */
    ea += dpp0_f<0x112>(ea); ba += dpp0_f<0x112>(ba);
    ea += dpp0_f<0x114>(ea); ba += dpp0_f<0x114>(ba);
    ea += dpp0_f<0x118>(ea); ba += dpp0_f<0x118>(ba);
    const float en = e_base + ea;
    dst[sub] = mu * __builtin_amdgcn_rcpf(en + 0.000000119209289f);
    dst[16 + sub] = b_base + ba;
    dst[32 + sub] = en;
  }

  /* one 128-sample block (NR:66-80).  ring: [2][128] floats in LDS holding this
   * instance's input, half `hc` = current block, the other half = previous block.
   * OUT_E: the block's output is e (notch) instead of y.  scr: [2][SCR] floats. */
  template <bool OUT_E>
  __device__ __forceinline__ void block(const float *ring, int hc, bool first, float mu, float *out,
                                        float *scr, int sub) {
    const int cb = hc * RDSP_BLOCK;
    const float *cur = ring + cb;
    const float *prv = ring + (cb ^ RDSP_BLOCK);
    const float *dsrc = first ? cur : prv; /* NR:69-79: first call d = x, then previous block */
    /* B_{-1} = X_{-2}.X_{-1} from the delay line as it stands */
    float bb = 0.f;
#pragma unroll
    for (int k = 0; k < TPL; k++) {
      const int a = TPL * sub + k;
      bb = fmaf(ring[(cb - 1 - a) & 255], ring[(cb - 2 - a) & 255], bb);
    }
    float b_base = row_allsum(bb);
    float e_base = energy;
    prepare(ring, cb, 0, sub, mu, e_base, b_base, scr);
    /* prologue: shift x_0 in; pp = this lane's part of A_0 = W_0.X_0 */
    xp[0] = row_shift_in(cur[0], xp[TPL & M]);
    float pp;
    {
      float q0 = w[0] * xp[0], q1 = w[1] * xp[1];
      q0 = fmaf(w[2], xp[2], q0); q1 = fmaf(w[3], xp[3], q1);
      q0 = fmaf(w[4], xp[4], q0); q1 = fmaf(w[5], xp[5], q1);
      pp = q0 + q1;
    }
    float g = 0.f; /* g_{-1}: no update pending */
#pragma unroll 1
    for (int s0 = 0; s0 < RDSP_BLOCK; s0 += 16) {
      const float *sc = scr + ((s0 >> 4) & 1) * SCR;
      __syncthreads(); /* the group's scalars are in LDS */
      float gi[16], bn[16], dd[16], in[20];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float4 a = *reinterpret_cast<const float4 *>(sc + 4 * q);
        float4 b = *reinterpret_cast<const float4 *>(sc + 16 + 4 * q);
        float4 c = *reinterpret_cast<const float4 *>(dsrc + s0 + 4 * q);
        gi[4 * q] = a.x; gi[4 * q + 1] = a.y; gi[4 * q + 2] = a.z; gi[4 * q + 3] = a.w;
        bn[4 * q] = b.x; bn[4 * q + 1] = b.y; bn[4 * q + 2] = b.z; bn[4 * q + 3] = b.w;
        dd[4 * q] = c.x; dd[4 * q + 1] = c.y; dd[4 * q + 2] = c.z; dd[4 * q + 3] = c.w;
      }
#pragma unroll
      for (int q = 0; q < 5; q++) { /* x_{s0} .. x_{s0+19}; past the block end the quad wraps, unused */
        float4 a = *reinterpret_cast<const float4 *>(ring + ((cb + s0 + 4 * q) & 255));
        in[4 * q] = a.x; in[4 * q + 1] = a.y; in[4 * q + 2] = a.z; in[4 * q + 3] = a.w;
      }
      e_base = sc[32 + 15];
      b_base = sc[16 + 15];
      if (s0 + 16 < RDSP_BLOCK) /* next group's scalars, written to the other half of scr */
        prepare(ring, cb, s0 + 16, sub, mu, e_base, b_base, scr + (((s0 >> 4) + 1) & 1) * SCR);
#pragma unroll
      for (int s = 0; s < 16; s++) {
        /* slot of step n = s0 + s.  On entry: g = g_{n-1}, w = W_{n-1}, pp = lane part of
         * A_n = W_{n-1}.X_n; ring: X_n[k] at xp[(wp + k) & M], X_{n-1}[k] one further. */
        const int wp = (-s) & M; /* s0 % 16 == 0 and NPH divides 16: compile-time */
        float r = pp + red_partner<0>(pp); /* reduction of A_n, stage 1 */
        w[0] = fmaf(g, xp[(wp + 1) & M], w[0]); /* W_n = W_{n-1} + g_{n-1} X_{n-1} */
        w[1] = fmaf(g, xp[(wp + 2) & M], w[1]);
        w[2] = fmaf(g, xp[(wp + 3) & M], w[2]);
        r += red_partner<1>(r);
        w[3] = fmaf(g, xp[(wp + 4) & M], w[3]);
        w[4] = fmaf(g, xp[(wp + 5) & M], w[4]);
        w[5] = fmaf(g, xp[(wp + 6) & M], w[5]);
        r += red_partner<2>(r);
        /* lane part of A_{n+1} = W_n.X_{n+1}; X_{n+1}[k] = X_n[k-1] for k >= 1 */
        float q1 = w[1] * xp[wp];
        float q0 = w[2] * xp[(wp + 1) & M];
        q1 = fmaf(w[3], xp[(wp + 2) & M], q1);
        q0 = fmaf(w[4], xp[(wp + 3) & M], q0);
        const float A = r + red_partner<3>(r);
        q1 = fmaf(w[5], xp[(wp + 4) & M], q1);
        float xnew = 0.f;
        if (s < 15 || s0 < RDSP_BLOCK - 16) { /* x_{n+1} exists: shift it in */
          xnew = row_shift_in(in[s + 1], xp[(wp + 5) & M]);
          xp[(wp + 7) & M] = xnew;
        }
        const float y = fmaf(g, bn[s], A);
        q0 = fmaf(w[0], xnew, q0);
        const float e = dd[s] - y;
        pp = q0 + q1;
        g = e * gi[s];
        out[s0 + s] = OUT_E ? e : y; /* every lane of the row holds the same value */
      }
    }
    /* the update of the last step is still pending: W_128 = W_127 + g_127 X_127 */
#pragma unroll
    for (int k = 0; k < TPL; k++) w[k] = fmaf(g, xp[(1 + k) & M], w[k]);
    energy = e_base;
  }
};

/* DUAL: both NLMS instances active (DSP-NR feeding the ALS filter).  The sketch never
 * enables both (CTL:240-296), so the common case keeps one instance in registers. */
template <bool DUAL>
__global__ void __launch_bounds__(64) rdsp_tail_shift_kernel(RdspTailParams p) {
  constexpr int CPW = 64 / LPC;
  constexpr int SPL = RDSP_BLOCK / LPC; /* samples per lane per block */
  constexpr int RINGS = DUAL ? 2 : 1;
  constexpr int PER_CH = (2 * RINGS + 1) * RDSP_BLOCK + 2 * SCR;
  __shared__ __attribute__((aligned(16))) float lds[CPW][PER_CH];
  if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x;
  const int sub = lane % LPC;
  const int cw = lane / LPC;
  size_t ch = (size_t)p.ch_base + (size_t)blockIdx.x * CPW + cw;
  const bool valid = ch < (size_t)p.n_channels;
  if (!valid) ch = p.n_channels - 1; /* compute on a real channel, store nothing */

  float *ringA = &lds[cw][0];                                   /* [2][128] kernel input         */
  float *ringB = DUAL ? &lds[cw][2 * RDSP_BLOCK] : ringA;       /* [2][128] ALS input when DUAL  */
  float *fin = &lds[cw][2 * RINGS * RDSP_BLOCK];                /* [128] final audio of the block */
  float *scr = &lds[cw][(2 * RINGS + 1) * RDSP_BLOCK];          /* [2][SCR] per-group scalars     */

  /* single-instance launches run whichever instance is on through `one`; with neither
   * on (SAM channels without noise reduction) the kernel is AGC + gain + pack only */
  const bool has_inst = DUAL || p.nr_on || p.als_mode;
  const bool one_is_nr = !DUAL && p.nr_on;
  float *o_w = one_is_nr ? p.nr_w : p.als_w;
  float *o_prev = one_is_nr ? p.nr_prev : p.als_prev;
  float *o_energy = one_is_nr ? p.nr_energy : p.als_energy;
  const float o_mu = one_is_nr ? p.nr_mu : p.als_mu;
  const int o_first = one_is_nr ? p.nr_first : p.als_first;
  const int o_mode = one_is_nr ? p.nr_mode : p.als_mode; /* 0: 1.1*y, 1: e, 2: y */

  Nlms nr, als; /* !DUAL: `als` is the one instance, `nr` stays dead */
  if constexpr (DUAL) {
    nr.load(p.nr_w, p.nr_prev, p.nr_energy, ch, sub);
    als.load(p.als_w, p.als_prev, p.als_energy, ch, sub);
  } else if (has_inst) {
    als.load(o_w, o_prev, o_energy, ch, sub);
  }
  float agc_g = p.st_scal[ch * 4 + 1];

  /* previous-block halves (half 1, since block 0 uses half 0 as current) */
  if constexpr (DUAL) {
#pragma unroll
    for (int k = 0; k < SPL; k++) {
      ringA[RDSP_BLOCK + sub * SPL + k] = p.nr_prev[ch * RDSP_BLOCK + sub * SPL + k];
      ringB[RDSP_BLOCK + sub * SPL + k] = p.als_prev[ch * RDSP_BLOCK + sub * SPL + k];
    }
  } else if (has_inst) {
#pragma unroll
    for (int k = 0; k < SPL; k++) ringA[RDSP_BLOCK + sub * SPL + k] = o_prev[ch * RDSP_BLOCK + sub * SPL + k];
  }

  const float *src = p.mid + ch * p.mid_stride;
  float4 nxa = *reinterpret_cast<const float4 *>(src + sub * SPL);
  float4 nxb = *reinterpret_cast<const float4 *>(src + sub * SPL + 4);

#pragma unroll 1
  for (int b = 0; b < p.n_blocks; b++) {
    const int hc = b & 1;
    *reinterpret_cast<float4 *>(ringA + hc * RDSP_BLOCK + sub * SPL) = nxa;
    *reinterpret_cast<float4 *>(ringA + hc * RDSP_BLOCK + sub * SPL + 4) = nxb;
    if (b + 1 < p.n_blocks) { /* next block's input lands while this block computes */
      nxa = *reinterpret_cast<const float4 *>(src + (size_t)(b + 1) * RDSP_BLOCK + sub * SPL);
      nxb = *reinterpret_cast<const float4 *>(src + (size_t)(b + 1) * RDSP_BLOCK + sub * SPL + 4);
    }
    __syncthreads();
    if constexpr (DUAL) { /* CONV:326-337, then the ALS filter */
      float *o = ringB + hc * RDSP_BLOCK;
      nr.template block<false>(ringA, hc, p.nr_first && b == 0, p.nr_mu, o, scr, sub);
      __syncthreads();
      if (p.nr_mode == 0) { /* CONV:334 */
#pragma unroll
        for (int k = 0; k < SPL; k++) o[sub * SPL + k] = mul_1p1(o[sub * SPL + k]);
        __syncthreads();
      }
      if (p.als_mode == 1) als.template block<true>(ringB, hc, p.als_first && b == 0, p.als_mu, fin, scr, sub);
      else als.template block<false>(ringB, hc, p.als_first && b == 0, p.als_mu, fin, scr, sub);
    } else if (has_inst) {
      if (o_mode == 1) als.template block<true>(ringA, hc, o_first && b == 0, o_mu, fin, scr, sub);
      else als.template block<false>(ringA, hc, o_first && b == 0, o_mu, fin, scr, sub);
    } else {
      *reinterpret_cast<float4 *>(fin + sub * SPL) = *reinterpret_cast<const float4 *>(ringA + hc * RDSP_BLOCK + sub * SPL);
      *reinterpret_cast<float4 *>(fin + sub * SPL + 4) = *reinterpret_cast<const float4 *>(ringA + hc * RDSP_BLOCK + sub * SPL + 4);
    }
    __syncthreads();
    /* A9 AGC + output gain + A10 pack: lane handles SPL consecutive samples */
    float L[SPL];
#pragma unroll
    for (int k = 0; k < SPL / 4; k++) {
      float4 a = *reinterpret_cast<const float4 *>(fin + sub * SPL + 4 * k);
      L[4 * k] = a.x; L[4 * k + 1] = a.y; L[4 * k + 2] = a.z; L[4 * k + 3] = a.w;
    }
    if (!DUAL && has_inst && o_mode == 0) { /* CONV:334 */
#pragma unroll
      for (int k = 0; k < SPL; k++) L[k] = mul_1p1(L[k]);
    }
    if (p.raw_out) { /* LMS_NoiseReduction(n, nrbuffer) in isolation, NR:66 */
      if (valid) {
#pragma unroll
        for (int k = 0; k < SPL; k++)
          p.raw_out[ch * p.mid_stride + (size_t)b * RDSP_BLOCK + sub * SPL + k] = L[k];
      }
      __syncthreads();
      continue;
    }
    if (p.agc_on) {
      float pw = 0.f;
#pragma unroll
      for (int k = 0; k < SPL; k++) pw += L[k] * L[k] + L[k] * L[k];
      pw = row_allsum(pw);
      float pp = pw / (float)(2 * RDSP_BLOCK);
      float rms = __builtin_amdgcn_sqrtf(pp); /* 1 ulp; the loop gain is a contraction */
      float gt = fminf(0.25f * __builtin_amdgcn_rcpf(rms + 1e-6f), 100.0f);
      float coef = (gt < agc_g) ? p.agc_attack : p.agc_decay;
      float gn = agc_g + coef * (gt - agc_g);
#pragma unroll
      for (int k = 0; k < SPL; k++) {
        int i = sub * SPL + k;
        float g = agc_g + (gn - agc_g) * ((float)(i + 1) / (float)RDSP_BLOCK);
        L[k] *= g;
      }
      agc_g = gn;
    }
    if (valid) {
      size_t o = ch * p.out_stride + (size_t)b * RDSP_BLOCK + sub * SPL;
#pragma unroll
      for (int k = 0; k < SPL; k += 4) {
        uint4 wv;
        float l0 = L[k] * p.out_gain, l1 = L[k + 1] * p.out_gain, l2 = L[k + 2] * p.out_gain,
              l3 = L[k + 3] * p.out_gain;
        wv.x = pack_lr(l0, l0); wv.y = pack_lr(l1, l1); wv.z = pack_lr(l2, l2); wv.w = pack_lr(l3, l3);
        *reinterpret_cast<uint4 *>(p.out_i16 + o + k) = wv;
        if (p.out_f32) {
          p.out_f32[o + k] = make_float2(l0, l0);
          p.out_f32[o + k + 1] = make_float2(l1, l1);
          p.out_f32[o + k + 2] = make_float2(l2, l2);
          p.out_f32[o + k + 3] = make_float2(l3, l3);
        }
      }
    }
    __syncthreads(); /* fin / rings are rewritten by the next block */
  }

  /* state out: weights, energy, last input block of each instance, AGC gain */
  if (valid) {
    const int hl = (p.n_blocks - 1) & 1;
    if constexpr (DUAL) {
      nr.store(p.nr_w, p.nr_energy, ch, sub);
      als.store(p.als_w, p.als_energy, ch, sub);
#pragma unroll
      for (int k = 0; k < SPL; k++) {
        p.nr_prev[ch * RDSP_BLOCK + sub * SPL + k] = ringA[hl * RDSP_BLOCK + sub * SPL + k];
        p.als_prev[ch * RDSP_BLOCK + sub * SPL + k] = ringB[hl * RDSP_BLOCK + sub * SPL + k];
      }
    } else if (has_inst) {
      als.store(o_w, o_energy, ch, sub);
#pragma unroll
      for (int k = 0; k < SPL; k++) o_prev[ch * RDSP_BLOCK + sub * SPL + k] = ringA[hl * RDSP_BLOCK + sub * SPL + k];
    }
    if (sub == 0 && !p.raw_out) p.st_scal[ch * 4 + 1] = agc_g;
  }
}

}  // namespace

/* the first tail kernel of round 1 (delay line shifted between lanes by DPP): experimental build only */
extern "C" int rdsp_launch_tail_shift(const RdspTailParams *p, hipStream_t stream) {
  const int grid = (p->n_channels - p->ch_base + 3) / 4;
  if (p->nr_on && p->als_mode) hipLaunchKernelGGL((rdsp_tail_shift_kernel<true>), dim3(grid), dim3(64), 0, stream, *p);
  else hipLaunchKernelGGL((rdsp_tail_shift_kernel<false>), dim3(grid), dim3(64), 0, stream, *p);
  return (int)hipGetLastError();
}
