/*
 * experimental/rdsp_tail_layouts.hip -- tail-kernel forms that were measured and not adopted
 * (EXPERIMENTAL=1 builds only; the product's kernel is ../rdsp_tail.hip; DESIGN.md 4.2, docs/history.md):
 *
 *   rdsp_tail1_kernel / rdsp_tail_x_kernel     one reduction per step (round 1); other lane layouts
 *       A7  NLMS noise reduction      RDSP_noise_reduction.h:35-80
 *       A8  ALS notch / peak          (AudioSDR, build-defined on A7's core)
 *       A9  AGC, output gain, A10 pack
 *
 * The NLMS recursion is serial in time: every step is a chain  dot product -> 16-lane
 * reduction -> error -> step size -> update, one wave per SIMD at 4096 channels.  Measured on
 * MI355X (tests/micro/dpp_kinds.hip, tail_bench.hip): a lone wave issues one instruction per
 * ~5 cycles whatever its kind (a DPP add 4.4, a plain fp32 op 3.0, a packed one 4.2 with three
 * waves on the SIMD), so a step costs its instruction count and its dependency chain.  The
 * kernel evaluates the recursion with a one-step lookahead,
 *        y_n = W_{n-1}.X_n + g_{n-1} (X_{n-1}.X_n) = A_n + g_{n-1} B_n,
 * which takes the reduction (A_n) off the g -> g chain (what stays loop-carried is fma, sub,
 * mul) and moves the energy E_n and the lag-1 correlation B_n, which depend on the input only,
 * into two DPP prefix scans per 64 steps.  Taps and delay line are <2 x float> values (packed
 * FMAs); the delay line is not shifted between lanes: every lane reads its next sample
 * x[n+1-TPL*sub] from the input ring in LDS (one ds_read_b32 per step and ring copy).
 *
 * Layouts.  The product is COLS = 16: a channel is one DPP row of 16 consecutive lanes, 6 taps
 * per lane, the reduction a 4-stage DPP butterfly.  An EXPERIMENTAL=1 build of the library also
 * carries the layouts that were measured and lost: half a row per channel
 * (COLS = 8) and, outside the north-star's "no MFMA", the 4 x 16 lane grid with the cross-lane
 * sums on the matrix pipe (COLS = 4 / 2: log2(COLS) quad_perm DPP steps, then one
 * v_mfma_f32_16x16x4_f32 with A = 1 sums the four rows of a column; a triangular A gives the
 * prefix sums).
 */
#include "rdsp_wave.h"

using namespace rdsp;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

/* per-group scalars in LDS: step size, B, energy for each of the GS steps of a group */

/* COLS = 16 is the third layout: a channel is one DPP row of 16 consecutive lanes and the
 * reduction is the 4-stage DPP butterfly (no matrix pipe); it shares everything else,
 * in particular the delay line fed from LDS instead of shifted by DPP.  COLS = 8 is the same
 * with half a row per channel: 12 taps per lane, 3 DPP stages, 8 channels per wave, i.e. half
 * the waves and 40 % fewer issue cycles per channel and step, which is what the front kernel
 * of the next call competes for. */
template <int COLS>
struct Geo {
  static_assert(COLS == 2 || COLS == 4 || COLS == 16 || COLS == 8, "2 or 4 columns of the lane grid, a whole row, or half a row");
  static constexpr bool ROW = COLS == 16 || COLS == 8; /* a channel is COLS consecutive lanes */
  static constexpr int LPC = ROW ? COLS : 4 * COLS;    /* lanes per channel */
  static constexpr int CPW = 64 / LPC;               /* channels per wave */
  static constexpr int TPL = RDSP_LMS_TAPS / LPC;    /* taps per lane: 12 or 6 */
  static constexpr int NPH = (TPL == 12) ? 16 : 8;   /* physical delay-line ring (>= TPL + 2, divides 128) */
  static constexpr int M = NPH - 1;
  static constexpr int SPL = RDSP_BLOCK / LPC;       /* samples per lane per block */
  static constexpr int NAC = (TPL == 12) ? 2 : 1;    /* packed accumulator chains of the dot product */
  /* steps per group: the lanes of a channel prepare a group's scalars together, GS / LPC
   * consecutive steps each.  The 16-lane row takes 32: its two prefix scans (8 DPP operations)
   * then serve 32 steps instead of 16 */
  static constexpr int GS = (COLS == 16) ? 32 : 16;
  static constexpr int SCR = 3 * GS;
};

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
/* sum over the lanes of a channel, result in all of them */
template <int COLS>
__device__ __forceinline__ float chan_sum(float v) {
  if constexpr (COLS == 16) return row_allsum(v);
  v += dpp_f<0xB1>(v);                          /* quad_perm [1,0,3,2] */
  if constexpr (COLS == 8) {
    v += dpp_f<0x4E>(v);                        /* quad_perm [2,3,0,1] */
    return v + dpp_f<0x141>(v);                 /* row_half_mirror */
  }
  if constexpr (COLS == 4) v += dpp_f<0x4E>(v); /* quad_perm [2,3,0,1] */
  return col_sum(v);
}
/* as dpp_f, lanes whose source falls outside the row read 0 (bound_ctrl) */
template <int CTRL>
__device__ __forceinline__ float dpp0_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

/* One NLMS instance of one channel.  Lane `sub` (= COLS*row + column inside the channel)
 * holds the taps of ages TPL*sub .. TPL*sub+TPL-1; CMSIS coefficient b[i] multiplies age 95-i. */
template <int COLS>
struct NlmsM {
  using G = Geo<COLS>;
  static constexpr int TPL = G::TPL, NPH = G::NPH, M = G::M, NAC = G::NAC, GS = G::GS, SCR = G::SCR;
  /* Taps and delay line as <2 x float> values so that the update and the dot product are
   * packed instructions (TPL/2 each instead of TPL; every VALU instruction costs a wave the
   * same 4 cycles).  w2[k] = (w[2k], w[2k+1]).  A step needs pairs of ring neighbours that
   * start at an even or an odd slot depending on the step's parity, and a packed operand
   * must be an aligned register pair, so the ring exists twice: xe[k] = (x[2k], x[2k+1]) and
   * xo[k] = (x[2k+1], x[2k+2]) (slots mod NPH).  The second copy costs no VALU work: the
   * lane's next sample is one more LDS read. */
  static_assert(TPL % 2 == 0 && NPH % 2 == 0, "taps pair up");
  v2f w2[TPL / 2];
  v2f xe[NPH / 2], xo[NPH / 2];
  float energy;

  /* the ring neighbours (x[i], x[i+1]); i is a compile-time constant wherever this is used */
  __device__ __forceinline__ v2f pair_at(int i) const {
    const int r = i & M;
    return (r & 1) ? xo[r >> 1] : xe[r >> 1];
  }
  __device__ __forceinline__ void put(int i, float v) { put(i, v, v); }
  /* ve and vo are the same sample read twice from LDS: a second read is LDS-pipe work, a copy
   * between the two rings would be one more VALU instruction per step */
  __device__ __forceinline__ void put(int i, float ve, float vo) {
    const int r = i & M, j = (i - 1) & M;
    xe[r >> 1][r & 1] = ve;
    xo[j >> 1][j & 1] = vo;
  }

  __device__ __forceinline__ void load(const float *wst, const float *prev, const float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) w2[t >> 1][t & 1] = wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))];
#pragma unroll
    for (int t = 0; t < NPH; t++) put(t, 0.f);
    /* before step s the in-lane tap t sits at physical ((-s) + 1 + t) & M */
#pragma unroll
    for (int t = 0; t < TPL; t++) put(t + 1, prev[ch * RDSP_BLOCK + (127 - (TPL * sub + t))]);
    energy = est[ch];
  }
  __device__ __forceinline__ void store(float *wst, float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))] = w2[t >> 1][t & 1];
    if (sub == 0) est[ch] = energy;
  }

  /* the lanes of a channel prepare the GS steps of a group (step order = lane order `sub`,
   * GS/LPC consecutive steps per lane): E_n, B_n by prefix sums of their increments, step size
   * mu/(E_n + eps).  ci = column inside the channel. */
  static __device__ __forceinline__ void prepare(const float *cur, int s0, int sub, int ci, float tri,
                                                 float mu, float e_base, float b_base, float *dst) {
    if constexpr (COLS == 2) {
      const float *x = cur + s0 + 2 * sub; /* the previous block sits right below the current one */
      const float xm = x[-1], x0 = x[0], x1 = x[1];
      const float qm = x[-97], q0 = x[-96], q1 = x[-95];
      const float ea0 = fmaf(x0, x0, -(q0 * q0)), ea1 = ea0 + fmaf(x1, x1, -(q1 * q1));
      const float ba0 = fmaf(x0, xm, -(q0 * qm)), ba1 = ba0 + fmaf(x1, x0, -(q1 * q0));
      /* exclusive offset of this lane: all lanes of the rows above, plus the even lane of its pair */
      const float se = ea1 + dpp_f<0xB1>(ea1), sb = ba1 + dpp_f<0xB1>(ba1);
      const float pe = col_prefix(se, tri), pb = col_prefix(sb, tri);
      const float oe = pe - (ci ? ea1 : se), ob = pb - (ci ? ba1 : sb);
      const float e0 = e_base + (oe + ea0), e1 = e_base + (oe + ea1);
      float2 *d2 = reinterpret_cast<float2 *>(dst);
      d2[sub] = make_float2(mu * __builtin_amdgcn_rcpf(e0 + 0.000000119209289f),
                            mu * __builtin_amdgcn_rcpf(e1 + 0.000000119209289f));
      d2[8 + sub] = make_float2(b_base + (ob + ba0), b_base + (ob + ba1));
      d2[16 + sub] = make_float2(e0, e1);
    } else if constexpr (COLS == 8) {
      /* two consecutive steps per lane; inclusive scan of the lane totals over the half row
       * (row_shr 1, 2, 4; a lane takes nothing from the neighbouring channel) */
      const float *x = cur + s0 + 2 * sub;
      const float xm = x[-1], x0 = x[0], x1 = x[1];
      const float qm = x[-97], q0 = x[-96], q1 = x[-95];
      const float ea0 = fmaf(x0, x0, -(q0 * q0)), ea1 = ea0 + fmaf(x1, x1, -(q1 * q1));
      const float ba0 = fmaf(x0, xm, -(q0 * qm)), ba1 = ba0 + fmaf(x1, x0, -(q1 * q0));
      float ie = ea1, ib = ba1;
      { const float te = dpp0_f<0x111>(ie), tb = dpp0_f<0x111>(ib); ie += (ci >= 1) ? te : 0.f; ib += (ci >= 1) ? tb : 0.f; }
      { const float te = dpp0_f<0x112>(ie), tb = dpp0_f<0x112>(ib); ie += (ci >= 2) ? te : 0.f; ib += (ci >= 2) ? tb : 0.f; }
      { const float te = dpp0_f<0x114>(ie), tb = dpp0_f<0x114>(ib); ie += (ci >= 4) ? te : 0.f; ib += (ci >= 4) ? tb : 0.f; }
      const float oe = ie - ea1, ob = ib - ba1; /* the lanes before this one */
      const float e0 = e_base + (oe + ea0), e1 = e_base + (oe + ea1);
      float2 *d2 = reinterpret_cast<float2 *>(dst);
      d2[sub] = make_float2(mu * __builtin_amdgcn_rcpf(e0 + 0.000000119209289f),
                            mu * __builtin_amdgcn_rcpf(e1 + 0.000000119209289f));
      d2[8 + sub] = make_float2(b_base + (ob + ba0), b_base + (ob + ba1));
      d2[16 + sub] = make_float2(e0, e1);
    } else if constexpr (COLS == 16) {
      /* two consecutive steps per lane (32 per group); inclusive scans of the lane totals over
       * the row: row_shr 1, 2, 4, 8 with zero fill */
      const float *x = cur + s0 + 2 * sub;
      const float xm = x[-1], x0 = x[0], x1 = x[1];
      const float qm = x[-97], q0 = x[-96], q1 = x[-95];
      const float ea0 = fmaf(x0, x0, -(q0 * q0)), ea1 = ea0 + fmaf(x1, x1, -(q1 * q1)); /* E_n - E_{n-1}, summed */
      const float ba0 = fmaf(x0, xm, -(q0 * qm)), ba1 = ba0 + fmaf(x1, x0, -(q1 * q0)); /* B_n - B_{n-1}, summed */
      float ie = ea1, ib = ba1;
      ie += dpp0_f<0x111>(ie); ib += dpp0_f<0x111>(ib);
      ie += dpp0_f<0x112>(ie); ib += dpp0_f<0x112>(ib);
      ie += dpp0_f<0x114>(ie); ib += dpp0_f<0x114>(ib);
      ie += dpp0_f<0x118>(ie); ib += dpp0_f<0x118>(ib);
      const float oe = ie - ea1, ob = ib - ba1; /* the lanes before this one */
      const float e0 = e_base + (oe + ea0), e1 = e_base + (oe + ea1);
      float2 *d2 = reinterpret_cast<float2 *>(dst);
      d2[sub] = make_float2(mu * __builtin_amdgcn_rcpf(e0 + 0.000000119209289f),
                            mu * __builtin_amdgcn_rcpf(e1 + 0.000000119209289f));
      d2[GS / 2 + sub] = make_float2(b_base + (ob + ba0), b_base + (ob + ba1));
      d2[GS + sub] = make_float2(e0, e1);
    } else {
      const float *x = cur + s0 + sub;
      const float xm = x[-1], x0 = x[0], qm = x[-97], q0 = x[-96];
      float ea = fmaf(x0, x0, -(q0 * q0)); /* E_n - E_{n-1} */
      float ba = fmaf(x0, xm, -(q0 * qm)); /* B_n - B_{n-1} */
      /* inclusive scan over the quad, then the rows above through the matrix pipe */
      const float e1 = dpp_f<0x90>(ea), b1 = dpp_f<0x90>(ba); /* quad_perm [0,0,1,2] */
      ea += (ci >= 1) ? e1 : 0.f;
      ba += (ci >= 1) ? b1 : 0.f;
      const float e2 = dpp_f<0x44>(ea), b2 = dpp_f<0x44>(ba); /* quad_perm [0,1,0,1] */
      ea += (ci >= 2) ? e2 : 0.f;
      ba += (ci >= 2) ? b2 : 0.f;
      const float se = dpp_f<0xFF>(ea), sb = dpp_f<0xFF>(ba); /* quad_perm [3,3,3,3]: the quad's sum */
      const float pe = col_prefix(se, tri), pb = col_prefix(sb, tri);
      const float en = e_base + ((pe - se) + ea);
      dst[sub] = mu * __builtin_amdgcn_rcpf(en + 0.000000119209289f);
      dst[16 + sub] = b_base + ((pb - sb) + ba);
      dst[32 + sub] = en;
    }
  }

  /* one 128-sample block; see Nlms::block in rdsp_tail.hip for the recursion.  ring is
   * [previous block | current block], 256 floats, so every sample a step looks back at
   * is at a fixed distance below it (no wrap) */
  template <bool OUT_E>
  __device__ __forceinline__ void block(const float *ring, bool first, float mu, float *out, float *scr,
                                        int sub, int ci, float tri) {
    const float *cur = ring + RDSP_BLOCK;
    const float *dsrc = first ? cur : ring; /* NR:69-79 */
    const float *mine = cur - TPL * sub;    /* this lane's newest tap of X_n is mine[n] */
    float bb = 0.f; /* B_{-1} = X_{-2}.X_{-1} */
#pragma unroll
    for (int t = 0; t < TPL; t++) bb = fmaf(mine[-1 - t], mine[-2 - t], bb);
    float b_base = chan_sum<COLS>(bb);
    float e_base = energy;
    prepare(cur, 0, sub, ci, tri, mu, e_base, b_base, scr);
    /* prologue: pp = lane part of A_0 = W_0.X_0 */
    put(0, mine[0]);
    float pp;
    {
      v2f q = w2[0] * pair_at(0);
#pragma unroll
      for (int k = 1; k < TPL / 2; k++) q = __builtin_elementwise_fma(w2[k], pair_at(2 * k), q);
      pp = q[0] + q[1];
    }
    float g = 0.f;
    /* x_{n+1} of the step about to run, read from LDS a whole step before its use -- twice, through
     * an offset the compiler cannot see through, so that each ring copy gets its own load */
    int zero = 0;
    asm volatile("" : "+v"(zero));
    const float *mine_b = mine + zero;
    float xn = mine[1], xnb = mine_b[1];
#pragma unroll 1
    for (int s0 = 0; s0 < RDSP_BLOCK; s0 += GS) {
      const float *sc = scr + ((s0 / GS) & 1) * SCR;
      __syncthreads();
      /* per-step scalars a quad of steps at a time (the loads of quad q+1 are issued before
       * the steps of quad q; sched_barrier keeps the compiler from hoisting a whole group
       * into registers); the lane's next sample is a one-dword LDS read per step */
      float4 gq = *reinterpret_cast<const float4 *>(sc);
      float4 bq = *reinterpret_cast<const float4 *>(sc + GS);
      float4 dq = *reinterpret_cast<const float4 *>(dsrc + s0);
      e_base = sc[2 * GS + GS - 1];
      b_base = sc[GS + GS - 1];
      if (s0 + GS < RDSP_BLOCK)
        prepare(cur, s0 + GS, sub, ci, tri, mu, e_base, b_base, scr + (((s0 / GS) + 1) & 1) * SCR);
#pragma unroll
      for (int q = 0; q < GS / 4; q++) {
        const float gi[4] = {gq.x, gq.y, gq.z, gq.w}, bn[4] = {bq.x, bq.y, bq.z, bq.w};
        const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
        if (q < GS / 4 - 1) {
          gq = *reinterpret_cast<const float4 *>(sc + 4 * (q + 1));
          bq = *reinterpret_cast<const float4 *>(sc + GS + 4 * (q + 1));
          dq = *reinterpret_cast<const float4 *>(dsrc + s0 + 4 * (q + 1));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int s = 4 * q + u;
          /* slot of step n = s0 + s.  On entry: g = g_{n-1}, w = W_{n-1}, pp = lane part of
           * A_n = W_{n-1}.X_n; ring: X_n[t] at xp[(wp + t) & M], X_{n-1}[t] one further. */
          const int wp = (-s) & M;
          const float A = chan_sum<COLS>(pp); /* needed only after the update below */
          /* the lane's next-but-one sample: the read is issued a step ahead, so its LDS latency
           * is not part of the recursion (it was: ~60 cycles of every step) */
          const bool more2 = s < GS - 2 || s0 < RDSP_BLOCK - GS; /* x_{n+2} exists */
          const float xn2 = more2 ? mine[s0 + s + 2] : 0.f;
          const float xn2b = more2 ? mine_b[s0 + s + 2] : 0.f;
          const float xnew = xn; /* 0 after the last sample of the block */
          const v2f gg = {g, g};
#pragma unroll
          for (int k = 0; k < TPL / 2; k++) w2[k] = __builtin_elementwise_fma(gg, pair_at(wp + 2 * k + 1), w2[k]); /* W_n */
          /* lane part of A_{n+1} = W_n.X_{n+1}; X_{n+1}[t] = X_n[t-1] for t >= 1 and X_{n+1}[0] =
           * x_{n+1}, which goes to the slot below X_n[0] (after step 127 nothing reads that slot
           * before the next block's first sample replaces it).  The pair that holds the new
           * sample comes last: its LDS read has the other products to land behind. */
          v2f acc[NAC];
#pragma unroll
          for (int k = 1; k < TPL / 2; k++) {
            const int a = (k - 1) % NAC;
            acc[a] = (k - 1 < NAC) ? w2[k] * pair_at(wp + 2 * k - 1)
                                   : __builtin_elementwise_fma(w2[k], pair_at(wp + 2 * k - 1), acc[a]);
          }
          const float y = fmaf(g, bn[u], A);
          const float e = dd[u] - y;
          put(wp + NPH - 1, xnew, xnb);
          xn = xn2;
          xnb = xn2b;
          acc[NAC - 1] = __builtin_elementwise_fma(w2[0], pair_at(wp - 1), acc[NAC - 1]);
          if constexpr (NAC == 2) acc[0] += acc[1];
          pp = acc[0][0] + acc[0][1];
          g = e * gi[u];
          out[s0 + s] = OUT_E ? e : y; /* every lane of the channel holds the same value */
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    {
      const v2f gg = {g, g};
#pragma unroll
      for (int k = 0; k < TPL / 2; k++) w2[k] = __builtin_elementwise_fma(gg, pair_at(1 + 2 * k), w2[k]); /* the pending update of step 127 */
    }
    energy = e_base;
  }
};

template <int COLS, bool DUAL, typename NL = NlmsM<COLS>>
__device__ __forceinline__ void tailm_body(const RdspTailParams &p) {
  using G = Geo<COLS>;
  constexpr int CPW = G::CPW, SPL = G::SPL;
  constexpr int RINGS = DUAL ? 2 : 1;
  /* +4: consecutive channels start four LDS banks apart */
  constexpr int PER_CH = (2 * RINGS + 1) * RDSP_BLOCK + 2 * NL::SCR + 4;
  __shared__ __attribute__((aligned(16))) float lds[CPW][PER_CH];
  if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x;
  const int row = lane >> 4, col = lane & 15;
  const int cw = G::ROW ? lane / G::LPC : col / COLS, ci = G::ROW ? lane % G::LPC : col % COLS;
  const int sub = G::ROW ? ci : COLS * row + ci;
  const float tri = (row <= (col >> 2)) ? 1.0f : 0.0f;
  size_t ch = (size_t)p.ch_base + (size_t)blockIdx.x * CPW + cw;
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

  NL nr, als; /* !DUAL: `als` is the one instance */
  if constexpr (DUAL) {
    nr.load(p.nr_w, p.nr_prev, p.nr_energy, ch, sub);
    als.load(p.als_w, p.als_prev, p.als_energy, ch, sub);
  } else if (has_inst) {
    als.load(o_w, o_prev, o_energy, ch, sub);
  }
  float agc_g = p.st_scal[ch * 4 + 1];

  /* the lower half of a ring is the previous block */
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
  const float4 *src4 = reinterpret_cast<const float4 *>(p.mid + ch * p.mid_stride + sub * SPL);

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
    { /* this block's input: not kept in registers through the step loop (that would cost the
         registers that let two front waves and this one share a SIMD; measured 3 % alone) */
      const float4 *n4 = src4 + (size_t)b * (RDSP_BLOCK / 4);
      float4 *dst4 = reinterpret_cast<float4 *>(ringA + RDSP_BLOCK + sub * SPL);
#pragma unroll
      for (int k = 0; k < SPL / 4; k++) dst4[k] = n4[k];
    }
    __syncthreads();
    if constexpr (DUAL) { /* CONV:326-337, then the ALS filter */
      float *o = ringB + RDSP_BLOCK;
      nr.template block<false>(ringA, p.nr_first && b == 0, p.nr_mu, o, scr, sub, ci, tri);
      __syncthreads();
      if (p.nr_mode == 0) { /* CONV:334 */
#pragma unroll
        for (int k = 0; k < SPL; k++) o[sub * SPL + k] = mul_1p1(o[sub * SPL + k]);
        __syncthreads();
      }
      if (p.als_mode == 1) als.template block<true>(ringB, p.als_first && b == 0, p.als_mu, fin, scr, sub, ci, tri);
      else als.template block<false>(ringB, p.als_first && b == 0, p.als_mu, fin, scr, sub, ci, tri);
    } else if (has_inst) {
      if (o_mode == 1) als.template block<true>(ringA, o_first && b == 0, o_mu, fin, scr, sub, ci, tri);
      else als.template block<false>(ringA, o_first && b == 0, o_mu, fin, scr, sub, ci, tri);
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
      pw = chan_sum<COLS>(pw);
      float pp = pw / (float)(2 * RDSP_BLOCK);
      float rms = __builtin_amdgcn_sqrtf(pp); /* 1 ulp; the loop gain is a contraction */
      float gt = fminf(0.25f * __builtin_amdgcn_rcpf(rms + 1e-6f), 100.0f);
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

  if (valid) { /* the last block processed is the upper half of the ring */
    if constexpr (DUAL) {
      nr.store(p.nr_w, p.nr_energy, ch, sub);
      als.store(p.als_w, p.als_energy, ch, sub);
#pragma unroll
      for (int k = 0; k < SPL; k++) {
        p.nr_prev[ch * RDSP_BLOCK + sub * SPL + k] = ringA[RDSP_BLOCK + sub * SPL + k];
        p.als_prev[ch * RDSP_BLOCK + sub * SPL + k] = ringB[RDSP_BLOCK + sub * SPL + k];
      }
    } else if (has_inst) {
      als.store(o_w, o_energy, ch, sub);
#pragma unroll
      for (int k = 0; k < SPL; k++) o_prev[ch * RDSP_BLOCK + sub * SPL + k] = ringA[RDSP_BLOCK + sub * SPL + k];
    }
    if (sub == 0 && !p.raw_out) p.st_scal[ch * 4 + 1] = agc_g;
  }
}

/* the one-step form of round 1 (one reduction per step), kept selectable for A/B runs */
__global__ void __launch_bounds__(64) rdsp_tail1_dual_kernel(RdspTailParams p) { tailm_body<16, true>(p); }
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(56))) rdsp_tail1_kernel(RdspTailParams p) {
  tailm_body<16, false>(p);
}

template <int COLS, bool DUAL>
__global__ void __launch_bounds__(64) rdsp_tail_x_kernel(RdspTailParams p) { tailm_body<COLS, DUAL>(p); }

template <int COLS>
int launch_x(const RdspTailParams *p, hipStream_t stream) {
  const int grid = (p->n_channels - p->ch_base + Geo<COLS>::CPW - 1) / Geo<COLS>::CPW;
  if (p->nr_on && p->als_mode) hipLaunchKernelGGL((rdsp_tail_x_kernel<COLS, true>), dim3(grid), dim3(64), 0, stream, *p);
  else hipLaunchKernelGGL((rdsp_tail_x_kernel<COLS, false>), dim3(grid), dim3(64), 0, stream, *p);
  return (int)hipGetLastError();
}

}  // namespace

/* variants of EXPERIMENTAL=1 builds: 102 one reduction per step (round 1), 101 half a row per
 * channel, 116 / 108 16 / 8 lanes with the reduction on the matrix pipe */
extern "C" int rdsp_launch_tail_layouts(const RdspTailParams *p, int variant, hipStream_t stream) {
  if (variant == 102) {
    const int grid = (p->n_channels - p->ch_base + 3) / 4;
    if (p->nr_on && p->als_mode) hipLaunchKernelGGL(rdsp_tail1_dual_kernel, dim3(grid), dim3(64), 0, stream, *p);
    else hipLaunchKernelGGL(rdsp_tail1_kernel, dim3(grid), dim3(64), 0, stream, *p);
    return (int)hipGetLastError();
  }
  if (variant == 101) return launch_x<8>(p, stream);
  if (variant == 116) return launch_x<4>(p, stream);
  if (variant == 108) return launch_x<2>(p, stream);
  return (int)hipErrorNotSupported;
}
