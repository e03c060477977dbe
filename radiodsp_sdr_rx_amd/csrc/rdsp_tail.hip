/*
 * rdsp_tail.hip -- the serial-in-time stages of the receive chain (gfx950):
 *
 *   rdsp_tail_kernel / rdsp_tail_dual_kernel   one channel per 16-lane DPP row, four per wave
 *       A7  NLMS noise reduction      RDSP_noise_reduction.h:35-80
 *       A8  ALS notch / peak          (AudioSDR, build-defined on A7's core)
 *       A9  AGC, output gain, A10 pack
 *
 * The NLMS recursion is serial in time and 4096 channels are 1024 waves: ONE wave per SIMD.
 * Measured for that regime (tests/micro/lone_wave.hip): a lone wave issues an instruction every
 * 5.1 cycles whatever its kind (6.5 for an LDS instruction), a dependent VALU instruction waits
 * 8.3, a DPP instruction that reads a fresh VALU result 16.4, an LDS round trip ~64 -- so a step
 * costs the larger of its instruction count x ~5.4 and its dependency chain.  NlmsB evaluates two
 * steps per 16-lane reduction with the energy E and the lag-1 correlation B, which depend on the
 * input only, in two DPP prefix scans per 64 steps: ~19 issue slots per step (103 cycles) against a
 * chain of ~98; measured 123 cycles per step.  Taking the chain away (weights one block stale,
 * hand-interleaved issue order: tests/micro/not_adopted/rdsp_tail_lookahead.h) was built and measured in round 3
 * and loses: its two extra scans cost more issue slots than the chain it removes.
 * Input blocks are fetched from HBM a whole block ahead (round 3: 0.967 -> 0.912 ms alone).
 */
#include "rdsp_wave.h"

using namespace rdsp;

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

/* as dpp_f, lanes whose source falls outside the row read 0 (bound_ctrl) */
template <int CTRL>
__device__ __forceinline__ float dpp0_f(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

/* The reduction of NlmsB: a0 / a1 are a lane's parts of A_0 / A_1.  Lanes 0-7 of the row end up
 * with A_0 (the sum of a0 over the 16 lanes), lanes 8-15 with A_1.  The first stage merges the two
 * values with write masks -- `v_add_f32_dpp ... bank_mask` writes the enabled banks only, the other
 * lanes keep the destination -- which the compiler's DPP combiner cannot express (it folds a
 * masked `update_dpp` into the add only when the disabled lanes may take an identity), so it is
 * written out, hazard wait states included: a VALU result read through DPP needs two wait states.
 * The hazard recognizer does not look into inline asm: the sources coming in get their two wait
 * states here (it adds at most one in front of an asm statement); for `t` going out it does treat
 * the statement as a VALU write and puts the two wait states before a DPP consumer itself
 * (checked over all instances in the generated code). */
__device__ __forceinline__ float reduce_halves(float a0, float a1) {
  float t;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(t)
      : "v"(a0), "v"(a1));
  return t;
}

/* ---- NlmsB: two steps per reduction, everything on one chain (round 2) ------------------------
 * With W the weights after update n0-1, two consecutive outputs are
 *     y_{n0}   = W.X_{n0}                         = A_0
 *     y_{n0+1} = W.X_{n0+1} + g_{n0} (X_{n0}.X_{n0+1}) = A_1 + g_{n0} B_{n0+1},
 * so the two dot products share the weights and are evaluated together: the accumulator is the
 * packed pair (A_0, A_1), every tap one v_pk_fma of the tap (broadcast to both halves by op_sel)
 * with the sample pair (x[m], x[m+1]) -- no horizontal add at the end -- and the 16-lane
 * reduction runs once per two steps: lanes 0-7 send their A_1 partial across (row_ror:8) and keep
 * A_0, lanes 8-15 the other way round, then three butterfly stages inside each half; the
 * consumers read A_0 / A_1 with row_newbcast.  Both tap updates use the same sample pairs with
 * the halves swapped (op_sel again).
 * Sample pairs live in an 8-slot register ring, Pair(m) = (mine[m], mine[m+1]) at slot m & 7
 * (mine[m] = x[m - 6 sub]: the newest of this lane's six samples of X_m); a block at n0 uses
 * Pair(n0-5 .. n0) and the two new ones are read one block ahead, into the slots of the two
 * pairs that went out of use two blocks earlier. */
struct NlmsB {
  /* 64 steps per group of scalars: the two prefix scans (8 DPP operations) serve four steps per lane */
  static constexpr int TPL = 6, GS = 64, SCR = 3 * GS;
  static constexpr int LDS_SCR = 2 * SCR;      /* double-buffered groups */
  static constexpr bool OUT_IN_SCR = false;    /* block() writes its 128 outputs to `out` */
  v2f w2[TPL / 2];
  v2f P[8];
  float energy;
  float emin; /* smallest energy + eps this lane has divided by (health word: <= 0 means a blow-up) */

  /* m odd: two dwords (the compiler pairs them as ds_read2_b32: 4 LDS cycles, and the two channels of a
   * 32-lane group -- lane addresses m - 6 sub, all of one parity -- collide whatever 16-byte-aligned
   * offset lies between them: 8).  m even: ONE aligned ds_read_b64 (2 LDS cycles; conflict-free with
   * the channels 32 dwords apart mod 64, see PER_CH) -- volatile, so that it is neither split into
   * dwords nor paired into a ds_read2_b64 (8 cycles).  In pipelined K3 the tail kernel's LDS cycles are
   * more than the front kernel's (PMC, round 3) and these reads were 60 % of them. */
  static __device__ __forceinline__ v2f pair_ld(const float *mine, int m) { return v2f{mine[m], mine[m + 1]}; }
  static __device__ __forceinline__ v2f pair_ld_even(const float *mine, int m) {
    typedef const volatile __attribute__((address_space(3))) v2f *LP;
    return *(LP)(mine + m);
  }
  __device__ __forceinline__ void load(const float *wst, const float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) w2[t >> 1][t & 1] = wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))];
    energy = est[ch];
    emin = __builtin_inff();
  }
  __device__ __forceinline__ void store(float *wst, float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))] = w2[t >> 1][t & 1];
    if (sub == 0) est[ch] = energy;
  }

  /* the 16 lanes of a channel prepare the 64 steps of a group, four consecutive steps each: E_n and
   * B_n by prefix sums of their increments (x_n^2 - x_{n-96}^2, x_n x_{n-1} - x_{n-96} x_{n-97}),
   * step size mu / (E_n + eps); dst = [step size x 64 | B x 64 | E x 64] */
  static __device__ __forceinline__ void prepare(const float *cur, int s0, int sub, float mu, float e_base,
                                                 float b_base, float *dst, float &emin) {
    const float *x = cur + s0 + 4 * sub; /* the previous block sits right below the current one */
    const float xm = x[-1], qm = x[-97];
    const float4 xv = *reinterpret_cast<const float4 *>(x), qv = *reinterpret_cast<const float4 *>(x - 96);
    const float xs[5] = {xm, xv.x, xv.y, xv.z, xv.w}, qs[5] = {qm, qv.x, qv.y, qv.z, qv.w};
    float ea[4], ba[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float de = fmaf(xs[k + 1], xs[k + 1], -(qs[k + 1] * qs[k + 1]));
      const float db = fmaf(xs[k + 1], xs[k], -(qs[k + 1] * qs[k]));
      ea[k] = (k == 0) ? de : ea[k - 1] + de;
      ba[k] = (k == 0) ? db : ba[k - 1] + db;
    }
    float ie = ea[3], ib = ba[3]; /* inclusive scans of the lane totals over the row: row_shr 1, 2, 4, 8 */
    ie += dpp0_f<0x111>(ie); ib += dpp0_f<0x111>(ib);
    ie += dpp0_f<0x112>(ie); ib += dpp0_f<0x112>(ib);
    ie += dpp0_f<0x114>(ie); ib += dpp0_f<0x114>(ib);
    ie += dpp0_f<0x118>(ie); ib += dpp0_f<0x118>(ib);
    const float oe = e_base + (ie - ea[3]), ob = b_base + (ib - ba[3]); /* everything before this lane */
    float e[4], g[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      e[k] = oe + ea[k];
      b[k] = ob + ba[k];
      const float den = e[k] + 0.000000119209289f;
      g[k] = mu * __builtin_amdgcn_rcpf(den);
      emin = fminf(emin, den);
    }
    float4 *d4 = reinterpret_cast<float4 *>(dst);
    d4[sub] = make_float4(g[0], g[1], g[2], g[3]);
    d4[GS / 4 + sub] = make_float4(b[0], b[1], b[2], b[3]);
    d4[GS / 2 + sub] = make_float4(e[0], e[1], e[2], e[3]);
  }

  /* one 128-sample block.  ring is [previous block | current block], 256 floats, so every sample a
   * step looks back at is at a fixed distance below it (no wrap) */
  template <bool OUT_E>
  __device__ __forceinline__ void block(const float *ring, bool first, float mu, float *out, float *scr, int sub, bool running) {
    const float *cur = ring + RDSP_BLOCK;
    const float *dsrc = first ? cur : ring; /* NR:69-79 */
    const float *mine = cur - TPL * sub;
    /* (the even pairs are read by instructions of their own: consecutive pairs overlap by a sample, and
     * left to itself the compiler reads the shared dword once and glues the pairs together with v_mov
     * instructions that wait for the LDS right behind the reads) */
    /* B_{-1} = X_{-2}.X_{-1} and E_{-1} = |X_{-1}|^2, both as exact sums over the 96-sample window in front
     * of the block.  arm_lms_norm_f32 carries the energy as a running difference for the whole stream
     * (`energy -= x0 * x0; energy += in * in`): after a loud-to-quiet transition what is left is the
     * rounding residue of everything that went through, of either sign and larger than the quiet window's
     * true energy, and `energy + 1.19e-7` at or below zero turns the step size negative or infinite.  The
     * prefix-scan form of the same sum (prepare) drew such residues MORE often than the sequential one
     * (round 3: 90 of 320 transitions flagged, 8 channels dead, against 1 in the CPU restatement).  Started
     * from the window sum at every block the residue lives for one block instead of for ever; between two
     * anchors it is the reference's difference form.  In ordinary signals both differ by the reference's own
     * accumulated rounding (~1e-6 relative). */
    float bb = 0.f, ee = 0.f;
#pragma unroll
    for (int t = 0; t < TPL; t++) {
      bb = fmaf(mine[-1 - t], mine[-2 - t], bb);
      ee = fmaf(mine[-1 - t], mine[-1 - t], ee);
    }
    float b_base = row_allsum(bb);
    float e_base = row_allsum(ee);
    /* rdsp_set_nlms_energy_mode(chain, 1): the reference's arithmetic -- the energy of the previous block's last
     * step carries on (`energy -= x0 * x0; energy += in * in` over the whole stream, NR:73 / arm_lms_norm_f32),
     * residue and all; B is this kernel's own quantity and stays the window sum */
    e_base = running ? energy : e_base;
    prepare(cur, 0, sub, mu, e_base, b_base, scr, emin);
#pragma unroll
    for (int m = -5; m <= 0; m++) P[m & 7] = (m & 1) ? pair_ld(mine, m) : pair_ld_even(mine, m);
#pragma unroll 1
    for (int s0 = 0; s0 < RDSP_BLOCK; s0 += GS) {
      const float *sc = scr + ((s0 / GS) & 1) * SCR;
      wg_sync<1>();
      float4 gq = *reinterpret_cast<const float4 *>(sc);
      float4 bq = *reinterpret_cast<const float4 *>(sc + GS);
      float4 dq = *reinterpret_cast<const float4 *>(dsrc + s0);
      e_base = sc[2 * GS + GS - 1];
      b_base = sc[GS + GS - 1];
      if (s0 + GS < RDSP_BLOCK) prepare(cur, s0 + GS, sub, mu, e_base, b_base, scr + (((s0 / GS) + 1) & 1) * SCR, emin);
#pragma unroll
      for (int q = 0; q < GS / 4; q++) {
        const float gi[4] = {gq.x, gq.y, gq.z, gq.w}, bn[4] = {bq.x, bq.y, bq.z, bq.w};
        const float dd[4] = {dq.x, dq.y, dq.z, dq.w};
        if (q < GS / 4 - 1) {
          gq = *reinterpret_cast<const float4 *>(sc + 4 * (q + 1));
          bq = *reinterpret_cast<const float4 *>(sc + GS + 4 * (q + 1));
          dq = *reinterpret_cast<const float4 *>(dsrc + s0 + 4 * (q + 1));
        }
        float o4[4];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int n = 4 * q + 2 * h; /* block start inside the group; s0 is a multiple of 8 */
          /* the two pairs the block after this one adds go straight into the slots that died two
           * blocks ago (Pair(n-7), Pair(n-6)): a whole block of time for the LDS reads to land */
          if ((n + 3 < GS) || (s0 + GS < RDSP_BLOCK)) {
            P[(n + 1) & 7] = pair_ld(mine, s0 + n + 1);   /* odd: two dwords */
            P[(n + 2) & 7] = pair_ld_even(mine, s0 + n + 2);
          }
          /* (A_0, A_1) lane parts: taps k = 0..5 against Pair(n - k) */
          v2f acc = v2f{w2[0][0], w2[0][0]} * P[n & 7];
          acc = __builtin_elementwise_fma(v2f{w2[0][1], w2[0][1]}, P[(n - 1) & 7], acc);
          acc = __builtin_elementwise_fma(v2f{w2[1][0], w2[1][0]}, P[(n - 2) & 7], acc);
          acc = __builtin_elementwise_fma(v2f{w2[1][1], w2[1][1]}, P[(n - 3) & 7], acc);
          acc = __builtin_elementwise_fma(v2f{w2[2][0], w2[2][0]}, P[(n - 4) & 7], acc);
          acc = __builtin_elementwise_fma(v2f{w2[2][1], w2[2][1]}, P[(n - 5) & 7], acc);
          /* one reduction for both: lanes 0-7 take their partner's A_0 part, lanes 8-15 the A_1 part */
          const float t = reduce_halves(acc[0], acc[1]);
          const float dA0 = dd[2 * h] - dpp_f<0x150>(t);     /* row_newbcast:0 */
          const float dA1 = dd[2 * h + 1] - dpp_f<0x158>(t); /* row_newbcast:8 */
          const float g0 = dA0 * gi[2 * h];                  /* e_{n0} = d - A_0 */
          const float e1 = fmaf(-g0, bn[2 * h + 1], dA1);    /* d - (A_1 + g_{n0} B_{n0+1}) */
          const float g1 = e1 * gi[2 * h + 1];
          o4[2 * h] = OUT_E ? dA0 : dd[2 * h] - dA0;
          o4[2 * h + 1] = OUT_E ? e1 : dd[2 * h + 1] - e1;
          /* W += g_{n0} X_{n0} + g_{n0+1} X_{n0+1}: tap pair kk against the swapped sample pairs */
          const v2f gg0 = {g0, g0}, gg1 = {g1, g1};
#pragma unroll
          for (int kk = 0; kk < TPL / 2; kk++) {
            const v2f p0 = P[(n - 2 * kk - 1) & 7], p1 = P[(n - 2 * kk) & 7];
            w2[kk] = __builtin_elementwise_fma(gg0, __builtin_shufflevector(p0, p0, 1, 0), w2[kk]);
            w2[kk] = __builtin_elementwise_fma(gg1, __builtin_shufflevector(p1, p1, 1, 0), w2[kk]);
          }
        }
        *reinterpret_cast<float4 *>(out + s0 + 4 * q) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    energy = e_base;
  }
  /* output sample i of the block just processed */
  static __device__ __forceinline__ float4 out4(const float *out, const float *, int i) {
    return *reinterpret_cast<const float4 *>(out + i);
  }
};

#ifdef RDSP_EXPERIMENTAL
#include "rdsp_tail_lookahead.h" /* tests/micro/not_adopted/, on the include path of EXPERIMENTAL=1 builds */
#endif

/* health word of one NLMS instance at the end of a launch (kernel params, st_status): bit 0 when some
 * lane of the channel divided by energy + eps <= 0, bit 1 when a weight or the energy is not finite
 * (0 * x is NaN exactly for those; the row sum carries it to every lane) */
template <typename NL>
__device__ __forceinline__ uint32_t nlms_health(const NL &f) {
  float z = f.energy * 0.f;
#pragma unroll
  for (int k = 0; k < NL::TPL / 2; k++) z = fmaf(f.w2[k][0], 0.f, fmaf(f.w2[k][1], 0.f, z));
  const float bad_e = row_allsum(f.emin <= 0.f ? 1.f : 0.f);
  const float nf = row_allsum(z);
  return (bad_e > 0.f ? 1u : 0u) | (nf == 0.f ? 0u : 2u);
}

#ifdef RDSP_TAIL_PROFILE /* measurement builds only (tests/micro/tail_phases.sh): s_memtime around the phases */
#define RDSP_TP(i) do { const long long tp_now = __builtin_readcyclecounter(); tp_acc[i] += tp_now - tp_last; tp_last = tp_now; } while (0)
#else
#define RDSP_TP(i) do { } while (0)
#endif

/* One wave per workgroup throughout: wg_sync<1>() (rdsp_wave.h) orders the lanes' LDS traffic without the
 * `s_waitcnt lgkmcnt(0)` a __syncthreads() leaves behind. */
template <bool DUAL, typename NL>
__device__ __forceinline__ void tail_body(const RdspTailParams &p) {
#ifdef RDSP_TAIL_PROFILE
  long long tp_acc[4] = {0, 0, 0, 0}, tp_last = __builtin_readcyclecounter();
#endif
  constexpr int CPW = 4, SPL = RDSP_BLOCK / 16; /* channels per wave, samples per lane per block */
  constexpr int RINGS = DUAL ? 2 : 1;
  constexpr int FIN = NL::OUT_IN_SCR ? 0 : RDSP_BLOCK;
  /* consecutive channels start 32 dwords apart mod 64: the even sample pairs of the two channels of a
   * 32-lane group (float2 index -3 sub each) then fall on disjoint halves of the 64 banks */
  constexpr int PER_CH0 = 2 * RINGS * RDSP_BLOCK + FIN + NL::LDS_SCR;
  constexpr int PER_CH = PER_CH0 + (96 - PER_CH0 % 64) % 64;
  static_assert(PER_CH % 64 == 32 && PER_CH % 4 == 0, "channel pitch in LDS");
  static_assert(!(DUAL && NL::OUT_IN_SCR), "the two-instance kernel hands a block on through `out`");
  __shared__ __attribute__((aligned(16))) float lds[CPW][PER_CH];
  if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
  else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
  else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x;
  const int cw = lane >> 4, sub = lane & 15;
  size_t ch = (size_t)p.ch_base + (size_t)blockIdx.x * CPW + cw;
  const bool valid = ch < (size_t)p.n_channels;
  if (!valid) ch = p.n_channels - 1; /* compute on a real channel, store nothing */

  float *ringA = &lds[cw][0];
  float *ringB = DUAL ? &lds[cw][2 * RDSP_BLOCK] : ringA;
  float *fin = &lds[cw][2 * RINGS * RDSP_BLOCK];
  float *scr = &lds[cw][2 * RINGS * RDSP_BLOCK + FIN];

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
    nr.load(p.nr_w, p.nr_energy, ch, sub);
    als.load(p.als_w, p.als_energy, ch, sub);
  } else if (has_inst) {
    als.load(o_w, o_energy, ch, sub);
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
  static_assert(SPL == 8, "two float4 per lane and block");
  /* Input blocks are fetched from HBM a whole block ahead (a load stored to LDS right away waits out
   * the full memory latency, ~0.8 us per block: measured 10 % of the kernel) and go into the ring
   * right after the step loop, before the block's own global stores are issued, so that the wait
   * for the fetch never waits for those */
  float4 nx0 = src4[0], nx1 = src4[1];
  /* every load of the set-up has landed before the first look-ahead fetch goes out: the waits the
   * compiler then puts into the block loop are for the fetches only (with the set-up's loads still
   * counted as pending at the loop header it waited for the newest fetch at the top of every block) */
  __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) */
  if (p.n_blocks > 0) {
    float4 *dst4 = reinterpret_cast<float4 *>(ringA + RDSP_BLOCK + sub * SPL);
    dst4[0] = nx0;
    dst4[1] = nx1;
    if (p.n_blocks > 1) {
      nx0 = src4[RDSP_BLOCK / 4];
      nx1 = src4[RDSP_BLOCK / 4 + 1];
    }
  }

#pragma unroll 1
  for (int b = 0; b < p.n_blocks; b++) {
    wg_sync<1>();
    RDSP_TP(0);
    if constexpr (DUAL) { /* CONV:326-337, then the ALS filter */
      float *o = ringB + RDSP_BLOCK;
      nr.template block<false>(ringA, p.nr_first && b == 0, p.nr_mu, o, scr, sub, p.energy_running != 0);
      wg_sync<1>();
      if (p.nr_mode == 0) { /* CONV:334 */
#pragma unroll
        for (int k = 0; k < SPL; k++) o[sub * SPL + k] = mul_1p1(o[sub * SPL + k]);
        wg_sync<1>();
      }
      if (p.als_mode == 1) als.template block<true>(ringB, p.als_first && b == 0, p.als_mu, fin, scr, sub, p.energy_running != 0);
      else als.template block<false>(ringB, p.als_first && b == 0, p.als_mu, fin, scr, sub, p.energy_running != 0);
    } else if (has_inst) {
      if (o_mode == 1) als.template block<true>(ringA, o_first && b == 0, o_mu, fin, scr, sub, p.energy_running != 0);
      else als.template block<false>(ringA, o_first && b == 0, o_mu, fin, scr, sub, p.energy_running != 0);
    }
    wg_sync<1>();
    RDSP_TP(1);
    /* A9 AGC + output gain + A10 pack: lane handles SPL consecutive samples */
    float L[SPL];
#pragma unroll
    for (int k = 0; k < SPL / 4; k++) {
      const float4 a = has_inst ? NL::out4(fin, scr, sub * SPL + 4 * k)
                                : *reinterpret_cast<const float4 *>(ringA + RDSP_BLOCK + sub * SPL + 4 * k);
      L[4 * k] = a.x; L[4 * k + 1] = a.y; L[4 * k + 2] = a.z; L[4 * k + 3] = a.w;
    }
    if (!DUAL && has_inst && o_mode == 0) { /* CONV:334 */
#pragma unroll
      for (int k = 0; k < SPL; k++) L[k] = mul_1p1(L[k]);
    }
    if (b + 1 < p.n_blocks) { /* the block just processed becomes the previous one, the next one moves in */
      float4 *r4 = reinterpret_cast<float4 *>(ringA + sub * SPL);
#pragma unroll
      for (int k = 0; k < SPL / 4; k++) r4[k] = r4[RDSP_BLOCK / 4 + k];
      if constexpr (DUAL) {
        float4 *q4 = reinterpret_cast<float4 *>(ringB + sub * SPL);
#pragma unroll
        for (int k = 0; k < SPL / 4; k++) q4[k] = q4[RDSP_BLOCK / 4 + k];
      }
      r4[RDSP_BLOCK / 4] = nx0;
      r4[RDSP_BLOCK / 4 + 1] = nx1;
      if (b + 2 < p.n_blocks) {
        const float4 *n4 = src4 + (size_t)(b + 2) * (RDSP_BLOCK / 4);
        nx0 = n4[0];
        nx1 = n4[1];
      }
    }
    RDSP_TP(2);
    if (p.raw_out) { /* LMS_NoiseReduction(n, nrbuffer) in isolation, NR:66 */
      if (valid) {
#pragma unroll
        for (int k = 0; k < SPL; k++)
          p.raw_out[ch * p.mid_stride + (size_t)b * RDSP_BLOCK + sub * SPL + k] = L[k];
      }
      wg_sync<1>();
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
    wg_sync<1>();
    RDSP_TP(3);
  }
#ifdef RDSP_TAIL_PROFILE
  if (p.out_f32 == nullptr && blockIdx.x == 0 && threadIdx.x == 0) /* phases: top of block | NLMS block() | ring rotation | AGC, pack, stores */
    printf("tail phases (cycles per 128-step block): top %lld  nlms %lld  rotate %lld  agc/pack %lld\n", tp_acc[0] / p.n_blocks,
           tp_acc[1] / p.n_blocks, tp_acc[2] / p.n_blocks, tp_acc[3] / p.n_blocks);
#endif

  if (p.st_status && has_inst) { /* sticky health words (rdsp_chain_get_status) */
    uint32_t h_nr = 0, h_als = 0;
    if constexpr (DUAL) {
      h_nr = nlms_health(nr);
      h_als = nlms_health(als);
    } else if (one_is_nr) {
      h_nr = nlms_health(als);
    } else {
      h_als = nlms_health(als);
    }
    if (valid && sub == 0) {
      if (h_nr) p.st_status[ch] |= h_nr;
      if (h_als) p.st_status[p.st_status_stride + ch] |= h_als;
    }
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

/* two NLMS instances (DSP-NR feeding the ALS filter; the sketch's menu never enables both, CTL:240-296) */
__global__ void __launch_bounds__(64) rdsp_tail_dual_kernel(RdspTailParams p) { tail_body<true, NlmsB>(p); }

/* The default (one NLMS instance).  124 VGPRs, 128 allocated: in pipelined mode it shares a SIMD's 512 with
 * two waves of the frequency-domain front kernel (176 allocated each: 480 in all); a tail wave that does
 * not fit waits for a front wave to retire (measured in round 1: 1.8 -> 2.4 ms per K3 step).  The CPU suite
 * reads both counts out of the built code object (test_generated_code_keeps_...). */
__global__ void __launch_bounds__(64) rdsp_tail_kernel(RdspTailParams p) {
  tail_body<false, NlmsB>(p);
}
#ifdef RDSP_EXPERIMENTAL
/* weights one block stale / four steps per reduction (tests/micro/not_adopted/rdsp_tail_lookahead.h): measured, not adopted */
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(64))) rdsp_tail_lookahead_kernel(RdspTailParams p) {
  tail_body<false, NlmsL>(p);
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(64))) rdsp_tail_four_kernel(RdspTailParams p) {
  tail_body<false, NlmsQ>(p);
}
#endif

}  // namespace

#ifdef RDSP_EXPERIMENTAL
extern "C" int rdsp_launch_tail_shift(const RdspTailParams *p, hipStream_t stream);            /* tests/micro/not_adopted/rdsp_tail_shift.hip */
extern "C" int rdsp_launch_tail_layouts(const RdspTailParams *p, int variant, hipStream_t stream); /* tests/micro/not_adopted/rdsp_tail_layouts.hip */
#endif

/* variant 100: the product's kernel.  EXPERIMENTAL=1 builds also know 104 (weights one block stale), 105 (four
 * steps per reduction),
 * 16 (delay line shifted by DPP), 102 (one reduction per step), 101 (half a row per channel),
 * 116 / 108 (16 / 8 lanes with the reduction on the matrix pipe). */
extern "C" int rdsp_launch_tail(const RdspTailParams *p, int variant, hipStream_t stream) {
  const int grid = (p->n_channels - p->ch_base + 3) / 4;
  const bool dual = p->nr_on && p->als_mode;
  if (variant == 100) {
    if (dual) hipLaunchKernelGGL(rdsp_tail_dual_kernel, dim3(grid), dim3(64), 0, stream, *p);
    else hipLaunchKernelGGL(rdsp_tail_kernel, dim3(grid), dim3(64), 0, stream, *p);
    return (int)hipGetLastError();
  }
#ifdef RDSP_EXPERIMENTAL
  if (variant == 104 || variant == 105) {
    if (dual) hipLaunchKernelGGL(rdsp_tail_dual_kernel, dim3(grid), dim3(64), 0, stream, *p);
    else if (variant == 104) hipLaunchKernelGGL(rdsp_tail_lookahead_kernel, dim3(grid), dim3(64), 0, stream, *p);
    else hipLaunchKernelGGL(rdsp_tail_four_kernel, dim3(grid), dim3(64), 0, stream, *p);
    return (int)hipGetLastError();
  }
  if (variant == 16) return rdsp_launch_tail_shift(p, stream);
  if (variant == 101 || variant == 102 || variant == 116 || variant == 108) return rdsp_launch_tail_layouts(p, variant, stream);
#endif
  return (int)hipErrorNotSupported;
}
