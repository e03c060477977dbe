/*
 * rdsp_tail8.hip -- the serial-in-time stages with eight lanes per channel and the
 * cross-lane sums on the matrix pipe (gfx950).  Same arithmetic as rdsp_tail.hip
 * (one-step lookahead NLMS, see there); what changes is where the cycles go.
 *
 * A DPP op occupies the VALU for ~11.6 cycles on MI355X (tests/micro/dpp_rate.hip), and
 * in the 16-lane kernel five of them per step (4-stage reduction + delay-line shift)
 * are half of its issue time -- time the front kernel of the next call wants, since
 * both kernels share the SIMDs in pipelined mode and the pair is VALU-bound.  Here
 *   * a channel is 8 lanes: columns j = 2c, 2c+1 of the 16x4 lane grid (lane = 16k + j),
 *     12 taps per lane, 8 channels per wave;
 *   * the reduction is one DPP (the column pair) plus one v_mfma_f32_16x16x4_f32
 *     with A = 1: D[i][j] = sum_k B[k][j] puts the sum over the four rows k of a column
 *     in every lane of that column (tests/micro/mfma_colsum.hip); fp32 products with
 *     1.0 are exact, the matrix pipe is otherwise idle, and the VALU is not involved;
 *   * the prefix sums of the per-group scalars use the same instruction with a
 *     triangular A (A[i][k] = k <= i/4);
 *   * the delay line is not shifted between lanes at all: every lane reads its next
 *     sample x[n+1-12*sub] from the input ring in LDS.
 * Per channel-step: ~1/8 DPP and ~4 plain VALU ops against 5/4 and ~4.5.
 */
#include "rdsp_wave.h"

using namespace rdsp;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int LPC = 8;                    /* lanes per channel */
constexpr int CPW = 64 / LPC;             /* channels per wave */
constexpr int TPL = RDSP_LMS_TAPS / LPC;  /* 12 taps per lane */
constexpr int NPH = 16;                   /* physical delay-line ring per lane (>= TPL + 2, divides 128) */
constexpr int M = NPH - 1;
constexpr int SCR = 48;                   /* per-group scalars: step size, B, energy */
constexpr int SPL = RDSP_BLOCK / LPC;     /* samples per lane per block */

/* sum over the four rows of a column: lanes j, j+16, j+32, j+48 -> every one of them */
__device__ __forceinline__ float col_sum(float v) {
  const v4f z = {0.f, 0.f, 0.f, 0.f};
  const v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, v, z, 0, 0, 0);
  return d[0];
}
/* inclusive prefix over the rows of a column; tri = (row <= (lane % 16) / 4) ? 1 : 0 */
__device__ __forceinline__ float col_prefix(float v, float tri) {
  const v4f z = {0.f, 0.f, 0.f, 0.f};
  const v4f d = __builtin_amdgcn_mfma_f32_16x16x4f32(tri, v, z, 0, 0, 0);
  return d[0];
}
__device__ __forceinline__ float pair_other(float v) { return dpp_f<0xB1>(v); } /* quad_perm [1,0,3,2] */
/* sum over the 8 lanes of a channel, result in all of them */
__device__ __forceinline__ float chan_sum(float v) { return col_sum(v + pair_other(v)); }

/* One NLMS instance of one channel.  Lane `sub` (= 2*row + column parity) holds the taps of
 * ages TPL*sub .. TPL*sub+TPL-1; CMSIS coefficient b[i] multiplies age 95-i. */
struct Nlms8 {
  float w[TPL];
  float xp[NPH];
  float energy;

  __device__ __forceinline__ void load(const float *wst, const float *prev, const float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) w[t] = wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))];
#pragma unroll
    for (int t = 0; t < NPH; t++) xp[t] = 0.f;
    /* before step s the in-lane tap t sits at physical ((-s) + 1 + t) & M */
#pragma unroll
    for (int t = 0; t < TPL; t++) xp[(t + 1) & M] = prev[ch * RDSP_BLOCK + (127 - (TPL * sub + t))];
    energy = est[ch];
  }
  __device__ __forceinline__ void store(float *wst, float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))] = w[t];
    if (sub == 0) est[ch] = energy;
  }

  /* lane `sub` prepares steps n0 = s0 + 2*sub and n0 + 1 of a 16-step group */
  static __device__ __forceinline__ void prepare(const float *cur, int s0, int sub, bool odd, float tri,
                                                 float mu, float e_base, float b_base, float *dst) {
    const float *x = cur + s0 + 2 * sub; /* the previous block sits right below the current one */
    const float xm = x[-1], x0 = x[0], x1 = x[1];
    const float qm = x[-97], q0 = x[-96], q1 = x[-95];
    const float ea0 = fmaf(x0, x0, -(q0 * q0)), ea1 = ea0 + fmaf(x1, x1, -(q1 * q1));
    const float ba0 = fmaf(x0, xm, -(q0 * qm)), ba1 = ba0 + fmaf(x1, x0, -(q1 * q0));
    /* exclusive offset of this lane: all lanes of the rows above, plus the even lane of its pair */
    const float se = ea1 + pair_other(ea1), sb = ba1 + pair_other(ba1);
    const float pe = col_prefix(se, tri), pb = col_prefix(sb, tri);
    const float oe = pe - (odd ? ea1 : se), ob = pb - (odd ? ba1 : sb);
    const float e0 = e_base + (oe + ea0), e1 = e_base + (oe + ea1);
    float2 *d2 = reinterpret_cast<float2 *>(dst);
    d2[sub] = make_float2(mu * __builtin_amdgcn_rcpf(e0 + 0.000000119209289f),
                          mu * __builtin_amdgcn_rcpf(e1 + 0.000000119209289f));
    d2[8 + sub] = make_float2(b_base + (ob + ba0), b_base + (ob + ba1));
    d2[16 + sub] = make_float2(e0, e1);
  }

  /* one 128-sample block; see Nlms::block in rdsp_tail.hip for the recursion.  ring is
   * [previous block | current block], 256 floats, so every sample a step looks back at
   * is at a fixed distance below it (no wrap) */
  template <bool OUT_E>
  __device__ __forceinline__ void block(const float *ring, bool first, float mu, float *out, float *scr,
                                        int sub, bool odd, float tri) {
    const float *cur = ring + RDSP_BLOCK;
    const float *dsrc = first ? cur : ring; /* NR:69-79 */
    const float *mine = cur - TPL * sub;    /* this lane's newest tap of X_n is mine[n] */
    float bb = 0.f; /* B_{-1} = X_{-2}.X_{-1} */
#pragma unroll
    for (int t = 0; t < TPL; t++) bb = fmaf(mine[-1 - t], mine[-2 - t], bb);
    float b_base = chan_sum(bb);
    float e_base = energy;
    prepare(cur, 0, sub, odd, tri, mu, e_base, b_base, scr);
    /* prologue: pp = lane part of A_0 = W_0.X_0 */
    xp[0] = mine[0];
    float pp;
    {
      float q0 = w[0] * xp[0], q1 = w[1] * xp[1], q2 = w[2] * xp[2];
#pragma unroll
      for (int t = 3; t < TPL; t += 3) {
        q0 = fmaf(w[t], xp[t], q0);
        q1 = fmaf(w[t + 1], xp[t + 1], q1);
        q2 = fmaf(w[t + 2], xp[t + 2], q2);
      }
      pp = (q0 + q1) + q2;
    }
    float g = 0.f;
#pragma unroll 1
    for (int s0 = 0; s0 < RDSP_BLOCK; s0 += 16) {
      const float *sc = scr + ((s0 >> 4) & 1) * SCR;
      __syncthreads();
      /* per-step scalars a quad of steps at a time (the loads of quad q+1 are issued before
       * the steps of quad q; sched_barrier keeps the compiler from hoisting a whole group
       * into registers); the lane's next sample is a one-dword LDS read per step */
      float4 gq = *reinterpret_cast<const float4 *>(sc);
      float4 bq = *reinterpret_cast<const float4 *>(sc + 16);
      float4 dq = *reinterpret_cast<const float4 *>(dsrc + s0);
      e_base = sc[32 + 15];
      b_base = sc[16 + 15];
      if (s0 + 16 < RDSP_BLOCK)
        prepare(cur, s0 + 16, sub, odd, tri, mu, e_base, b_base, scr + (((s0 >> 4) + 1) & 1) * SCR);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float gi[4] = {gq.x, gq.y, gq.z, gq.w}, bn[4] = {bq.x, bq.y, bq.z, bq.w};
        const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
        if (q < 3) {
          gq = *reinterpret_cast<const float4 *>(sc + 4 * (q + 1));
          bq = *reinterpret_cast<const float4 *>(sc + 16 + 4 * (q + 1));
          dq = *reinterpret_cast<const float4 *>(dsrc + s0 + 4 * (q + 1));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int s = 4 * q + u;
          /* slot of step n = s0 + s.  On entry: g = g_{n-1}, w = W_{n-1}, pp = lane part of
           * A_n = W_{n-1}.X_n; ring: X_n[t] at xp[(wp + t) & M], X_{n-1}[t] one further. */
          const int wp = (-s) & M;
          const float A = chan_sum(pp); /* one DPP, one MFMA; needed only after the update below */
          const bool more = s < 15 || s0 < RDSP_BLOCK - 16; /* x_{n+1} exists */
          const float xnew = more ? mine[s0 + s + 1] : 0.f;
#pragma unroll
          for (int t = 0; t < TPL; t++) w[t] = fmaf(g, xp[(wp + t + 1) & M], w[t]); /* W_n */
          /* lane part of A_{n+1} = W_n.X_{n+1}; X_{n+1}[t] = X_n[t-1] for t >= 1 */
          float q0 = w[1] * xp[wp], q1 = w[2] * xp[(wp + 1) & M], q2 = w[3] * xp[(wp + 2) & M];
#pragma unroll
          for (int t = 4; t + 2 < TPL; t += 3) {
            q0 = fmaf(w[t], xp[(wp + t - 1) & M], q0);
            q1 = fmaf(w[t + 1], xp[(wp + t) & M], q1);
            q2 = fmaf(w[t + 2], xp[(wp + t + 1) & M], q2);
          }
          q0 = fmaf(w[10], xp[(wp + 9) & M], q0);
          q1 = fmaf(w[11], xp[(wp + 10) & M], q1);
          const float y = fmaf(g, bn[u], A);
          const float e = dd[u] - y;
          if (more) xp[(wp + 15) & M] = xnew;
          q2 = fmaf(w[0], xnew, q2);
          pp = (q0 + q1) + q2;
          g = e * gi[u];
          out[s0 + s] = OUT_E ? e : y; /* every lane of the channel holds the same value */
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int t = 0; t < TPL; t++) w[t] = fmaf(g, xp[(1 + t) & M], w[t]); /* the pending update of step 127 */
    energy = e_base;
  }
};

template <bool DUAL>
__global__ void __launch_bounds__(64) rdsp_tail8_kernel(RdspTailParams p) {
  constexpr int RINGS = DUAL ? 2 : 1;
  /* +4: consecutive channels start four LDS banks apart (broadcast reads of 8 channels hit 32 banks) */
  constexpr int PER_CH = (2 * RINGS + 1) * RDSP_BLOCK + 2 * SCR + 4;
  __shared__ __attribute__((aligned(16))) float lds[CPW][PER_CH];
  const int lane = threadIdx.x;
  const int row = lane >> 4, col = lane & 15;
  const int cw = col >> 1;
  const bool odd = (col & 1) != 0;
  const int sub = 2 * row + (col & 1);
  const float tri = (row <= (col >> 2)) ? 1.0f : 0.0f;
  size_t ch = (size_t)blockIdx.x * CPW + cw;
  const bool valid = ch < (size_t)p.n_channels;
  if (!valid) ch = p.n_channels - 1; /* compute on a real channel, store nothing */

  float *ringA = &lds[cw][0];
  float *ringB = DUAL ? &lds[cw][2 * RDSP_BLOCK] : ringA;
  float *fin = &lds[cw][2 * RINGS * RDSP_BLOCK];
  float *scr = &lds[cw][(2 * RINGS + 1) * RDSP_BLOCK];

  const bool has_inst = DUAL || p.nr_on || p.als_mode;
  const bool one_is_nr = !DUAL && p.nr_on;
  float *o_w = one_is_nr ? p.nr_w : p.als_w;
  float *o_prev = one_is_nr ? p.nr_prev : p.als_prev;
  float *o_energy = one_is_nr ? p.nr_energy : p.als_energy;
  const float o_mu = one_is_nr ? p.nr_mu : p.als_mu;
  const int o_first = one_is_nr ? p.nr_first : p.als_first;
  const int o_mode = one_is_nr ? p.nr_mode : p.als_mode; /* 0: 1.1*y, 1: e, 2: y */

  Nlms8 nr, als; /* !DUAL: `als` is the one instance */
  if constexpr (DUAL) {
    nr.load(p.nr_w, p.nr_prev, p.nr_energy, ch, sub);
    als.load(p.als_w, p.als_prev, p.als_energy, ch, sub);
  } else if (has_inst) {
    als.load(o_w, o_prev, o_energy, ch, sub);
  }
  float agc_g = p.st_scal[ch * 4 + 1];

  if constexpr (DUAL) {
#pragma unroll
    for (int k = 0; k < SPL; k++) {
      ringA[sub * SPL + k] = p.nr_prev[ch * RDSP_BLOCK + sub * SPL + k];
      ringB[sub * SPL + k] = p.als_prev[ch * RDSP_BLOCK + sub * SPL + k];
    }
  } else if (has_inst) {
#pragma unroll
    for (int k = 0; k < SPL; k++) ringA[sub * SPL + k] = o_prev[ch * RDSP_BLOCK + sub * SPL + k];
  }

  const float *src = p.mid + ch * p.mid_stride;
  static_assert(SPL == 16, "four float4 per lane per block");
  const float4 *src4 = reinterpret_cast<const float4 *>(src + sub * SPL);
  float4 nxa = src4[0], nxb = src4[1], nxc = src4[2], nxd = src4[3];

#pragma unroll 1
  for (int b = 0; b < p.n_blocks; b++) {
    if (b > 0) { /* the block just processed becomes the previous one */
      float4 *r4 = reinterpret_cast<float4 *>(ringA + sub * SPL);
#pragma unroll
      for (int k = 0; k < SPL / 4; k++) r4[k] = r4[RDSP_BLOCK / 4 + k];
      if constexpr (DUAL) {
        float4 *q4 = reinterpret_cast<float4 *>(ringB + sub * SPL);
#pragma unroll
        for (int k = 0; k < SPL / 4; k++) q4[k] = q4[RDSP_BLOCK / 4 + k];
      }
    }
#ifndef RDSP_T8_PREFETCH
#define RDSP_T8_PREFETCH 0 /* 1: next block in registers through the step loop: +12 VGPRs, 3 % faster alone, but then two front waves + this one no longer fit a SIMD */
#endif
#if !RDSP_T8_PREFETCH
    if (b > 0) {
      const float4 *n4 = src4 + (size_t)b * (RDSP_BLOCK / 4);
      nxa = n4[0]; nxb = n4[1]; nxc = n4[2]; nxd = n4[3];
    }
#endif
    {
      float4 *dst4 = reinterpret_cast<float4 *>(ringA + RDSP_BLOCK + sub * SPL);
      dst4[0] = nxa; dst4[1] = nxb; dst4[2] = nxc; dst4[3] = nxd;
    }
#if RDSP_T8_PREFETCH
    if (b + 1 < p.n_blocks) { /* next block's input lands while this block computes */
      const float4 *n4 = src4 + (size_t)(b + 1) * (RDSP_BLOCK / 4);
      nxa = n4[0]; nxb = n4[1]; nxc = n4[2]; nxd = n4[3];
    }
#endif
    __syncthreads();
    if constexpr (DUAL) { /* CONV:326-337, then the ALS filter */
      float *o = ringB + RDSP_BLOCK;
      nr.template block<false>(ringA, p.nr_first && b == 0, p.nr_mu, o, scr, sub, odd, tri);
      __syncthreads();
      if (p.nr_mode == 0) { /* CONV:334 */
#pragma unroll
        for (int k = 0; k < SPL; k++) o[sub * SPL + k] *= 1.1f;
        __syncthreads();
      }
      if (p.als_mode == 1) als.template block<true>(ringB, p.als_first && b == 0, p.als_mu, fin, scr, sub, odd, tri);
      else als.template block<false>(ringB, p.als_first && b == 0, p.als_mu, fin, scr, sub, odd, tri);
    } else if (has_inst) {
      if (o_mode == 1) als.template block<true>(ringA, o_first && b == 0, o_mu, fin, scr, sub, odd, tri);
      else als.template block<false>(ringA, o_first && b == 0, o_mu, fin, scr, sub, odd, tri);
    } else {
#pragma unroll
      for (int k = 0; k < SPL / 4; k++)
        *reinterpret_cast<float4 *>(fin + sub * SPL + 4 * k) =
            *reinterpret_cast<const float4 *>(ringA + RDSP_BLOCK + sub * SPL + 4 * k);
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
      for (int k = 0; k < SPL; k++) L[k] *= 1.1f;
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
      pw = chan_sum(pw);
      float pp = pw / (float)(2 * RDSP_BLOCK);
      float rms = sqrtf(pp);
      float gt = fminf(0.25f / (rms + 1e-6f), 100.0f);
      float coef = (gt < agc_g) ? p.agc_attack : p.agc_decay;
      float gn = agc_g + coef * (gt - agc_g);
#pragma unroll
      for (int k = 0; k < SPL; k++) {
        int i = sub * SPL + k;
        float gg = agc_g + (gn - agc_g) * ((float)(i + 1) / (float)RDSP_BLOCK);
        L[k] *= gg;
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
    __syncthreads();
  }

  if (valid) {
    const int hl = 1; /* the last block processed is the upper half of the ring */
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

extern "C" int rdsp_launch_tail8(const RdspTailParams *p, hipStream_t stream) {
  const int grid = (p->n_channels + CPW - 1) / CPW;
  if (p->nr_on && p->als_mode) hipLaunchKernelGGL((rdsp_tail8_kernel<true>), dim3(grid), dim3(64), 0, stream, *p);
  else hipLaunchKernelGGL((rdsp_tail8_kernel<false>), dim3(grid), dim3(64), 0, stream, *p);
  return (int)hipGetLastError();
}
