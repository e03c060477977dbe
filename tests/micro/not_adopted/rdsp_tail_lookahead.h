/*
 * experimental/rdsp_tail_lookahead.h -- NlmsL: the NLMS recursion with the weights one block (two
 * steps) stale and a hand-interleaved issue order.  MEASURED AND NOT ADOPTED (round 3):
 * it takes the dot products, their reduction and the tap updates off the g -> g chain, and its
 * step loop does run faster (tests/micro/lone_wave.hip: 206 cycles per two steps against ~225), but a
 * lone wave issues one instruction per 5.1-6.5 cycles whatever it is, round 2's chain-bound kernel
 * already sits at 84 % of that issue bound, and the two extra prefix scans (lag-2 and lag-3
 * correlations) plus four more instructions per block cost more than the shorter chain saves:
 * 1.008 ms alone against 0.912 ms, pipelined K3 1.39 ms against 1.21 ms (same box).
 * Included by ../rdsp_tail.hip in EXPERIMENTAL=1 builds (variant 104).
 *
 * NlmsQ below (variant 105): four steps per reduction -- one transposing 16-lane reduction of depth 4
 * per four steps.  Its step loop is 21 % shorter than NlmsB's (tests/micro/lone_wave.hip: 355 cycles
 * per four steps against 2 x 225), but it needs the same four prefix scans as NlmsL, and those plus
 * the exact restart of the three correlations cost 2 900 cycles per 128-step block outside the loop
 * (s_memtime phases, tests/micro/tail_phases.sh): 0.927 ms alone against 0.901 ms, 2.39e8 VALU
 * instructions per launch against 2.18e8, pipelined K3 1.24 ms against 1.17 ms.  MEASURED, NOT ADOPTED.
 */
/* ---- NlmsL: weights one block stale, hand-interleaved issue order (the product) ---------------
 * Lane layout and sample pairs as NlmsB; the pair ring has 16 slots because a block's update runs
 * one block late: iteration n issues  RED(n) [of the dot products of block n, made an iteration
 * ago], UPD(n-2), DOT(n+2), REC(n)  and holds Pair(n-7 .. n+2) plus the two pairs in flight.
 * Per two steps the scratch carries one record of 8 floats written by prepare():
 *     { c_n, c_{n+1}, R1(n), R1(n+1) | R2(n), R2(n+1), R3(n+1), - },  c = mu / (E + eps);
 * the block's two outputs go back into words 6 and 7 of its record (read by the AGC / pack stage). */
struct NlmsL {
  static constexpr int TPL = 6, SCR = 8 * (RDSP_BLOCK / 2);
  static constexpr int LDS_SCR = SCR + 8;      /* + one record: the last iteration's look-ahead reads */
  static constexpr bool OUT_IN_SCR = true;
  v2f w2[TPL / 2];
  v2f P[16];
  float energy;
  float emin = __builtin_inff(); /* health word (see NlmsB): not tracked by the experimental forms */

  static __device__ __forceinline__ v2f pair_ld(const float *mine, int m) { return v2f{mine[m], mine[m + 1]}; }
  __device__ __forceinline__ void load(const float *wst, const float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) w2[t >> 1][t & 1] = wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))];
    energy = est[ch];
  }
  __device__ __forceinline__ void store(float *wst, float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))] = w2[t >> 1][t & 1];
    if (sub == 0) est[ch] = energy;
  }

  /* the 16 lanes of a channel prepare the 128 steps of a block, eight consecutive steps each:
   * E, R1, R2, R3 by prefix sums of their increments  x_n x_{n-L} - x_{n-96} x_{n-96-L}
   * (inside the lane, then one DPP scan of the lane totals over the row each).  Returns E_127. */
  static __device__ __forceinline__ float prepare(const float *cur, int sub, float mu, const float (&base)[4],
                                                  float *scr) {
    const float *x = cur + 8 * sub; /* the previous block sits right below the current one */
    float xs[11], qs[11];           /* xs[3 + k] = x[k], k = -3..7; qs likewise 96 samples earlier */
    {
      const float4 a = *reinterpret_cast<const float4 *>(x - 4), b = *reinterpret_cast<const float4 *>(x),
                   c = *reinterpret_cast<const float4 *>(x + 4);
      const float4 d = *reinterpret_cast<const float4 *>(x - 100), e = *reinterpret_cast<const float4 *>(x - 96),
                   f = *reinterpret_cast<const float4 *>(x - 92);
      xs[0] = a.y; xs[1] = a.z; xs[2] = a.w; xs[3] = b.x; xs[4] = b.y; xs[5] = b.z; xs[6] = b.w;
      xs[7] = c.x; xs[8] = c.y; xs[9] = c.z; xs[10] = c.w;
      qs[0] = d.y; qs[1] = d.z; qs[2] = d.w; qs[3] = e.x; qs[4] = e.y; qs[5] = e.z; qs[6] = e.w;
      qs[7] = f.x; qs[8] = f.y; qs[9] = f.z; qs[10] = f.w;
    }
    float val[4][8];
#pragma unroll
    for (int L = 0; L < 4; L++) {
      float a[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float d = fmaf(xs[3 + k], xs[3 + k - L], -(qs[3 + k] * qs[3 + k - L]));
        a[k] = (k == 0) ? d : a[k - 1] + d;
      }
      float in = a[7]; /* inclusive scan of the lane totals over the row: row_shr 1, 2, 4, 8 */
      in += dpp0_f<0x111>(in);
      in += dpp0_f<0x112>(in);
      in += dpp0_f<0x114>(in);
      in += dpp0_f<0x118>(in);
      const float off = base[L] + (in - a[7]); /* everything before this lane */
#pragma unroll
      for (int k = 0; k < 8; k++) val[L][k] = off + a[k];
    }
    float4 *rec = reinterpret_cast<float4 *>(scr) + 8 * sub; /* records 4 sub .. 4 sub + 3 */
#pragma unroll
    for (int b = 0; b < 4; b++) {
      const float c0 = mu * __builtin_amdgcn_rcpf(val[0][2 * b] + 0.000000119209289f);
      const float c1 = mu * __builtin_amdgcn_rcpf(val[0][2 * b + 1] + 0.000000119209289f);
      rec[2 * b] = make_float4(c0, c1, val[1][2 * b], val[1][2 * b + 1]);
      rec[2 * b + 1] = make_float4(val[2][2 * b], val[2][2 * b + 1], val[3][2 * b + 1], 0.f);
    }
    return dpp_f<0x15F>(val[0][7]); /* row_newbcast:15: E_127 to every lane of the channel */
  }

  template <bool OUT_E>
  __device__ __forceinline__ void block(const float *ring, bool first, float mu, float *, float *scr, int sub, bool = false) { /* (the energy mode is the product kernel's) */
    const float *cur = ring + RDSP_BLOCK;
    const float *dsrc = first ? cur : ring; /* NR:69-79 */
    const float *mine = cur - TPL * sub;
    int zero = 0;
    asm volatile("" : "+v"(zero));
    const float *mine_b = mine + zero;
    {
      /* R_L(-1) = X_{-1-L}.X_{-1} recomputed exactly at every block (only E is a running sum over
       * the whole stream, like the energy of arm_lms_norm_f32) */
      float m[9];
#pragma unroll
      for (int t = 0; t < 9; t++) m[t] = mine[-1 - t];
      float b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
      for (int t = 0; t < TPL; t++) {
        b1 = fmaf(m[t], m[t + 1], b1);
        b2 = fmaf(m[t], m[t + 2], b2);
        b3 = fmaf(m[t], m[t + 3], b3);
      }
      const float base[4] = {energy, row_allsum(b1), row_allsum(b2), row_allsum(b3)};
      energy = prepare(cur, sub, mu, base, scr);
    }
#pragma unroll
    for (int m = -7; m <= 2; m++) P[m & 15] = pair_ld(mine, m);
    __syncthreads();
    /* DOT(0) with the weights up to date: the first block has no corrections (g_{-2} = g_{-1} = 0
     * below; the matching UPD(-2) of iteration 0 adds 0 x finite samples) */
    v2f acc[2];
    acc[0] = v2f{w2[0][0], w2[0][0]} * P[0];
    acc[0] = __builtin_elementwise_fma(v2f{w2[0][1], w2[0][1]}, P[15], acc[0]);
    acc[0] = __builtin_elementwise_fma(v2f{w2[1][0], w2[1][0]}, P[14], acc[0]);
    acc[0] = __builtin_elementwise_fma(v2f{w2[1][1], w2[1][1]}, P[13], acc[0]);
    acc[0] = __builtin_elementwise_fma(v2f{w2[2][0], w2[2][0]}, P[12], acc[0]);
    acc[0] = __builtin_elementwise_fma(v2f{w2[2][1], w2[2][1]}, P[11], acc[0]);
    acc[1] = acc[0];
    asm volatile("s_nop 1"); /* RED reads acc through DPP: two wait states behind the VALU write */
    v2f G2 = {0.f, 0.f}, G1 = {0.f, 0.f}; /* (g_{n-2}, -), (g_{n-1}, -): the low halves are used */
    const float4 *rp = reinterpret_cast<const float4 *>(scr);
    float4 ra = rp[0], rb = rp[1];
    v2f dd = *reinterpret_cast<const v2f *>(dsrc);
#pragma unroll 1
    for (int c = 0; c < RDSP_BLOCK / 32; c++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int n = 2 * i;         /* block start mod 32 (ring slots repeat every 16 steps) */
        const int s = 32 * c + n;    /* block start inside the 128-sample block */
        v2f &aold = acc[i & 1], &anew = acc[(i + 1) & 1];
        const float c0 = ra.x, c1 = ra.y, r1a = ra.z, r1b = ra.w, r2a = rb.x, r2b = rb.y, r3b = rb.z;
        const float d0 = dd[0], d1 = dd[1];
        /* look-ahead reads: next record, next desired pair, the two sample pairs DOT(n+4) adds (into
         * the slots of Pair(n-13), Pair(n-12), long dead) */
        const float4 na = rp[(s >> 1) * 2 + 2], nb = rp[(s >> 1) * 2 + 3];
        const v2f nd = *reinterpret_cast<const v2f *>(dsrc + s + 2);
        P[(n + 3) & 15] = pair_ld(mine, s + 3);   /* odd: two dwords */
        P[(n + 4) & 15] = pair_ld(mine_b, s + 4); /* even: one ds_read_b64; through the opaque copy of the
                                                     pointer, or the compiler shares a dword of the two
                                                     reads and glues the pair with a v_mov that waits for
                                                     the LDS right here */
        float t, dA0, dA1, u0, u1, e0, e1, g0, g1;
        v2f acc2;
        /* Block A: RED(n) -- lanes 0-7 end up with A_n, lanes 8-15 with A_{n+1} -- interleaved with
         * UPD(n-2) and DOT(n+2).  Two asm statements instead of one per instruction: the hazard
         * recognizer counts no wait states across consecutive asm statements and puts an s_nop in
         * front of every statement that reads what an earlier one wrote. */
#define RDSP_UPD " op_sel:[0,1,0] op_sel_hi:[0,0,1]\n\t"
#define RDSP_ROW " row_mask:0xf bank_mask:0xf\n\t"
        asm volatile(
            "v_add_f32_dpp %[t], %[a0], %[a0] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_pk_fma_f32 %[w0], %[G2], %[pm3], %[w0]" RDSP_UPD
            "v_add_f32_dpp %[t], %[a1], %[a1] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_pk_fma_f32 %[w1], %[G2], %[pm5], %[w1]" RDSP_UPD
            "v_pk_fma_f32 %[w2], %[G2], %[pm7], %[w2]" RDSP_UPD
            "v_add_f32_dpp %[t], %[t], %[t] row_half_mirror" RDSP_ROW
            "v_pk_fma_f32 %[w0], %[G1], %[pm2], %[w0]" RDSP_UPD
            "v_pk_fma_f32 %[w1], %[G1], %[pm4], %[w1]" RDSP_UPD
            "v_pk_fma_f32 %[w2], %[G1], %[pm6], %[w2]" RDSP_UPD
            "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1]" RDSP_ROW
            "v_pk_mul_f32 %[an], %[w0], %[pp2] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 %[ac2], %[w0], %[pp1] op_sel:[1,0]\n\t"
            "v_pk_fma_f32 %[an], %[w1], %[p0], %[an] op_sel_hi:[0,1,1]\n\t"
            "v_add_f32_dpp %[t], %[t], %[t] quad_perm:[1,0,3,2]" RDSP_ROW
            "v_pk_fma_f32 %[ac2], %[w1], %[pm1], %[ac2] op_sel:[1,0,0]\n\t"
            "v_pk_fma_f32 %[an], %[w2], %[pm2], %[an] op_sel_hi:[0,1,1]"
            : [t] "=&v"(t), [w0] "+v"(w2[0]), [w1] "+v"(w2[1]), [w2] "+v"(w2[2]), [an] "=&v"(anew), [ac2] "=&v"(acc2)
            : [a0] "v"(aold[0]), [a1] "v"(aold[1]), [G2] "v"(G2), [G1] "v"(G1), [pm7] "v"(P[(n - 7) & 15]),
              [pm6] "v"(P[(n - 6) & 15]), [pm5] "v"(P[(n - 5) & 15]), [pm4] "v"(P[(n - 4) & 15]),
              [pm3] "v"(P[(n - 3) & 15]), [pm2] "v"(P[(n - 2) & 15]), [pm1] "v"(P[(n - 1) & 15]), [p0] "v"(P[n & 15]),
              [pp1] "v"(P[(n + 1) & 15]), [pp2] "v"(P[(n + 2) & 15]));
        /* Block B: REC(n), the end of DOT(n+2) as filler between its dependent links */
        asm volatile(
            "v_subrev_f32_dpp %[dA0], %[t], %[d0] row_newbcast:0" RDSP_ROW
            "v_subrev_f32_dpp %[dA1], %[t], %[d1] row_newbcast:8" RDSP_ROW
            "v_fma_f32 %[u0], -%[g2], %[r2a], %[dA0]\n\t"
            "v_fma_f32 %[u1], -%[g2], %[r3b], %[dA1]\n\t"
            "v_fma_f32 %[e0], -%[g1], %[r1a], %[u0]\n\t"
            "v_fma_f32 %[u1], -%[g1], %[r2b], %[u1]\n\t"
            "v_mul_f32 %[g0o], %[e0], %[c0]\n\t"
            "v_pk_fma_f32 %[ac2], %[w2], %[pm3], %[ac2] op_sel:[1,0,0]\n\t"
            "v_fma_f32 %[e1], -%[g0o], %[r1b], %[u1]\n\t"
            "v_pk_add_f32 %[an], %[an], %[ac2]\n\t"
            "v_mul_f32 %[g1o], %[e1], %[c1]"
            : [dA0] "=&v"(dA0), [dA1] "=&v"(dA1), [u0] "=&v"(u0), [u1] "=&v"(u1), [e0] "=&v"(e0), [e1] "=&v"(e1),
              [g0o] "=&v"(g0), [g1o] "=&v"(g1), [ac2] "+v"(acc2), [an] "+v"(anew)
            : [t] "v"(t), [d0] "v"(d0), [d1] "v"(d1), [w2] "v"(w2[2]), [pm3] "v"(P[(n - 3) & 15]), [g2] "v"(G2[0]),
              [g1] "v"(G1[0]), [r2a] "v"(r2a), [r3b] "v"(r3b), [r1a] "v"(r1a), [r2b] "v"(r2b), [c0] "v"(c0),
              [r1b] "v"(r1b), [c1] "v"(c1));
#undef RDSP_UPD
#undef RDSP_ROW
        /* outputs into words 6, 7 of this block's record */
        *reinterpret_cast<v2f *>(scr + 4 * s + 6) = OUT_E ? v2f{e0, e1} : v2f{d0 - e0, d1 - e1};
        {
          v2f ga, gb; /* only the low halves are read: the high halves stay undefined (no copies) */
          ga[0] = g0;
          gb[0] = g1;
          G2 = ga;
          G1 = gb;
        }
        ra = na; rb = nb; dd = nd;
      }
    }
    /* the pending update of the last block: UPD(126) */
    {
      const v2f gg0 = {G2[0], G2[0]}, gg1 = {G1[0], G1[0]};
#pragma unroll
      for (int kk = 0; kk < TPL / 2; kk++) {
        const v2f p0 = P[(126 - 2 * kk - 1) & 15], p1 = P[(126 - 2 * kk) & 15];
        w2[kk] = __builtin_elementwise_fma(gg0, __builtin_shufflevector(p0, p0, 1, 0), w2[kk]);
        w2[kk] = __builtin_elementwise_fma(gg1, __builtin_shufflevector(p1, p1, 1, 0), w2[kk]);
      }
    }
  }
  /* output samples i .. i+3 of the block just processed (i a multiple of 4) */
  static __device__ __forceinline__ float4 out4(const float *, const float *scr, int i) {
    const v2f a = *reinterpret_cast<const v2f *>(scr + 4 * i + 6), b = *reinterpret_cast<const v2f *>(scr + 4 * i + 14);
    return make_float4(a[0], a[1], b[0], b[1]);
  }
};


/* ---- NlmsQ: four steps per reduction ----------------------------------------------------------
 * With W the weights after update n-1, four consecutive outputs are
 *     y_n     = W.X_n
 *     y_{n+1} = W.X_{n+1} + g_n R1(n+1)
 *     y_{n+2} = W.X_{n+2} + g_n R2(n+2) + g_{n+1} R1(n+2)
 *     y_{n+3} = W.X_{n+3} + g_n R3(n+3) + g_{n+1} R2(n+3) + g_{n+2} R1(n+3),     R_L(m) = X_{m-L}.X_m,
 * so the four dot products share the weights: two packed accumulators (A_n, A_{n+1}), (A_{n+2},
 * A_{n+3}) and ONE transposing 16-lane reduction of depth 4 per four steps (8 DPP adds: lanes 0-3
 * end up with A_n, 4-7 with A_{n+2}, 8-11 with A_{n+1}, 12-15 with A_{n+3}) where NlmsB runs its
 * 5-deep one every two steps -- a dependent DPP costs a lone wave 16.4 cycles, and the reduction
 * was 40 % of NlmsB's dependency chain.  The energy E and the lag-1..3 correlations come from four
 * DPP prefix scans per 128 steps (prepare).  Per four steps: 12 + 8 + 4 + 10 + 12 = 46 VALU
 * instructions against 2 x 25.5, 8 LDS instructions against 8.6. */
__device__ __forceinline__ float reduce_quarters(v2f a, v2f b) {
  float ta, tb, u;
  asm("s_nop 1\n\t"
      "v_add_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %2, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t" /* lanes 0-3, 8-11 take lane + 4 */
      "v_add_f32_dpp %2, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"  /* lanes 4-7, 12-15 take lane - 4 */
      "s_nop 1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
      : "=&v"(ta), "=&v"(tb), "=&v"(u)
      : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
  return u;
}

struct NlmsQ {
  static constexpr int TPL = 6, REC = 12;              /* floats per four-step record */
  static constexpr int SCR = REC * (RDSP_BLOCK / 4);   /* one 128-step block of records */
  static constexpr int LDS_SCR = SCR + REC;            /* + one record: the last block's look-ahead reads */
  static constexpr bool OUT_IN_SCR = false;
  v2f w2[TPL / 2];
  v2f P[16]; /* Pair(m) = (mine[m], mine[m+1]) at slot m & 15; a block at n uses Pair(n-5 .. n+2) */
  float energy;
  float emin = __builtin_inff(); /* health word (see NlmsB): not tracked by the experimental forms */

  static __device__ __forceinline__ v2f pair_ld(const float *mine, int m) { return v2f{mine[m], mine[m + 1]}; }
  __device__ __forceinline__ void load(const float *wst, const float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) w2[t >> 1][t & 1] = wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))];
    energy = est[ch];
  }
  __device__ __forceinline__ void store(float *wst, float *est, size_t ch, int sub) {
#pragma unroll
    for (int t = 0; t < TPL; t++) wst[ch * RDSP_LMS_TAPS + (95 - (TPL * sub + t))] = w2[t >> 1][t & 1];
    if (sub == 0) est[ch] = energy;
  }

  /* the 16 lanes of a channel prepare the 128 steps of a block, eight consecutive steps (two
   * records) each: E, R1, R2, R3 by prefix sums of their increments  x_n x_{n-L} - x_{n-96} x_{n-96-L}
   * (inside the lane, then one DPP scan of the lane totals over the row each).  Record of the block
   * at n:  { c_n .. c_{n+3} | R1(n+1) R1(n+2) R1(n+3) R2(n+2) | R2(n+3) R3(n+3) - - },  c = mu / (E + eps).
   * Returns E_127. */
  static __device__ __forceinline__ float prepare(const float *cur, int sub, float mu, const float (&base)[4],
                                                  float *scr) {
    const float *x = cur + 8 * sub; /* the previous block sits right below the current one */
    float xs[11], qs[11];           /* xs[3 + k] = x[k], k = -3..7; qs likewise 96 samples earlier */
    {
      const float4 a = *reinterpret_cast<const float4 *>(x - 4), b = *reinterpret_cast<const float4 *>(x),
                   c = *reinterpret_cast<const float4 *>(x + 4);
      const float4 d = *reinterpret_cast<const float4 *>(x - 100), e = *reinterpret_cast<const float4 *>(x - 96),
                   f = *reinterpret_cast<const float4 *>(x - 92);
      xs[0] = a.y; xs[1] = a.z; xs[2] = a.w; xs[3] = b.x; xs[4] = b.y; xs[5] = b.z; xs[6] = b.w;
      xs[7] = c.x; xs[8] = c.y; xs[9] = c.z; xs[10] = c.w;
      qs[0] = d.y; qs[1] = d.z; qs[2] = d.w; qs[3] = e.x; qs[4] = e.y; qs[5] = e.z; qs[6] = e.w;
      qs[7] = f.x; qs[8] = f.y; qs[9] = f.z; qs[10] = f.w;
    }
    float val[4][8];
#pragma unroll
    for (int L = 0; L < 4; L++) {
      float a[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const float d = fmaf(xs[3 + k], xs[3 + k - L], -(qs[3 + k] * qs[3 + k - L]));
        a[k] = (k == 0) ? d : a[k - 1] + d;
      }
      float in = a[7]; /* inclusive scan of the lane totals over the row: row_shr 1, 2, 4, 8 */
      in += dpp0_f<0x111>(in);
      in += dpp0_f<0x112>(in);
      in += dpp0_f<0x114>(in);
      in += dpp0_f<0x118>(in);
      const float off = base[L] + (in - a[7]); /* everything before this lane */
#pragma unroll
      for (int k = 0; k < 8; k++) val[L][k] = off + a[k];
    }
    float4 *rec = reinterpret_cast<float4 *>(scr) + 6 * sub; /* records 2 sub, 2 sub + 1 */
#pragma unroll
    for (int b = 0; b < 2; b++) {
      float c[4];
#pragma unroll
      for (int k = 0; k < 4; k++) c[k] = mu * __builtin_amdgcn_rcpf(val[0][4 * b + k] + 0.000000119209289f);
      rec[3 * b] = make_float4(c[0], c[1], c[2], c[3]);
      rec[3 * b + 1] = make_float4(val[1][4 * b + 1], val[1][4 * b + 2], val[1][4 * b + 3], val[2][4 * b + 2]);
      rec[3 * b + 2] = make_float4(val[2][4 * b + 3], val[3][4 * b + 3], 0.f, 0.f);
    }
    return dpp_f<0x15F>(val[0][7]); /* row_newbcast:15: E_127 to every lane of the channel */
  }

  template <bool OUT_E>
  __device__ __forceinline__ void block(const float *ring, bool first, float mu, float *out, float *scr, int sub, bool = false) {
    const float *cur = ring + RDSP_BLOCK;
    const float *dsrc = first ? cur : ring; /* NR:69-79 */
    const float *mine = cur - TPL * sub;
    int zero = 0;
    asm volatile("" : "+v"(zero));
    const float *mine_b = mine + zero; /* an opaque copy: the even pairs are read through it, or the
                                          compiler shares dwords between overlapping pair reads and
                                          glues the pairs with v_mov (VALU work on the chain) */
    {
      /* R_L(-1) = X_{-1-L}.X_{-1} recomputed exactly at every block (only E is a running sum over
       * the whole stream, like the energy of arm_lms_norm_f32) */
      float m[9];
#pragma unroll
      for (int t = 0; t < 9; t++) m[t] = mine[-1 - t];
      float b1 = 0.f, b2 = 0.f, b3 = 0.f;
#pragma unroll
      for (int t = 0; t < TPL; t++) {
        b1 = fmaf(m[t], m[t + 1], b1);
        b2 = fmaf(m[t], m[t + 2], b2);
        b3 = fmaf(m[t], m[t + 3], b3);
      }
      const float base[4] = {energy, row_allsum(b1), row_allsum(b2), row_allsum(b3)};
      energy = prepare(cur, sub, mu, base, scr);
    }
#pragma unroll
    for (int m = -5; m <= 2; m++) P[m & 15] = pair_ld((m & 1) ? mine : mine_b, m);
    __syncthreads();
    const float4 *rp = reinterpret_cast<const float4 *>(scr);
    float4 ra = rp[0], rb = rp[1], rc = rp[2];
    float4 dq = *reinterpret_cast<const float4 *>(dsrc);
#pragma unroll 1
    for (int c = 0; c < RDSP_BLOCK / 16; c++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int n = 4 * i;      /* block start mod 16 (the ring slots repeat every 16 steps) */
        const int s = 16 * c + n; /* block start inside the 128-sample block */
        const float cc[4] = {ra.x, ra.y, ra.z, ra.w}, dd[4] = {dq.x, dq.y, dq.z, dq.w};
        const float r1_1 = rb.x, r1_2 = rb.y, r1_3 = rb.z, r2_2 = rb.w, r2_3 = rc.x, r3_3 = rc.y;
        /* look-ahead reads: the next record and desired samples, the four pairs the next block adds
         * (into the slots of Pair(n-13 .. n-10), long dead) */
        const float4 na = rp[3 * (s >> 2) + 3], nb = rp[3 * (s >> 2) + 4], nc = rp[3 * (s >> 2) + 5];
        const float4 nd = *reinterpret_cast<const float4 *>(dsrc + s + 4);
        P[(n + 3) & 15] = pair_ld(mine, s + 3);
        P[(n + 4) & 15] = pair_ld(mine_b, s + 4);
        P[(n + 5) & 15] = pair_ld(mine, s + 5);
        P[(n + 6) & 15] = pair_ld(mine_b, s + 6);
        /* (A_n, A_{n+1}) and (A_{n+2}, A_{n+3}) lane parts: taps k = 0..5 against Pair(n - k), Pair(n + 2 - k) */
        v2f accA = v2f{w2[0][0], w2[0][0]} * P[n & 15];
        v2f accB = v2f{w2[0][0], w2[0][0]} * P[(n + 2) & 15];
        accA = __builtin_elementwise_fma(v2f{w2[0][1], w2[0][1]}, P[(n - 1) & 15], accA);
        accB = __builtin_elementwise_fma(v2f{w2[0][1], w2[0][1]}, P[(n + 1) & 15], accB);
        accA = __builtin_elementwise_fma(v2f{w2[1][0], w2[1][0]}, P[(n - 2) & 15], accA);
        accB = __builtin_elementwise_fma(v2f{w2[1][0], w2[1][0]}, P[n & 15], accB);
        accA = __builtin_elementwise_fma(v2f{w2[1][1], w2[1][1]}, P[(n - 3) & 15], accA);
        accB = __builtin_elementwise_fma(v2f{w2[1][1], w2[1][1]}, P[(n - 1) & 15], accB);
        accA = __builtin_elementwise_fma(v2f{w2[2][0], w2[2][0]}, P[(n - 4) & 15], accA);
        accB = __builtin_elementwise_fma(v2f{w2[2][0], w2[2][0]}, P[(n - 2) & 15], accB);
        accA = __builtin_elementwise_fma(v2f{w2[2][1], w2[2][1]}, P[(n - 5) & 15], accA);
        accB = __builtin_elementwise_fma(v2f{w2[2][1], w2[2][1]}, P[(n - 3) & 15], accB);
        const float u = reduce_quarters(accA, accB);
        const float dA0 = dd[0] - dpp_f<0x150>(u); /* row_newbcast:0  */
        const float dA1 = dd[1] - dpp_f<0x158>(u); /* row_newbcast:8  */
        const float dA2 = dd[2] - dpp_f<0x154>(u); /* row_newbcast:4  */
        const float dA3 = dd[3] - dpp_f<0x15C>(u); /* row_newbcast:12 */
        const float g0 = dA0 * cc[0];
        const float e1 = fmaf(-g0, r1_1, dA1);
        const float t2 = fmaf(-g0, r2_2, dA2);
        const float t3 = fmaf(-g0, r3_3, dA3);
        const float g1 = e1 * cc[1];
        const float e2 = fmaf(-g1, r1_2, t2);
        const float t3b = fmaf(-g1, r2_3, t3);
        const float g2 = e2 * cc[2];
        const float e3 = fmaf(-g2, r1_3, t3b);
        const float g3 = e3 * cc[3];
        *reinterpret_cast<float4 *>(out + s) = OUT_E ? make_float4(dA0, e1, e2, e3)
                                                     : make_float4(dd[0] - dA0, dd[1] - e1, dd[2] - e2, dd[3] - e3);
        /* W += sum_j g_{n+j} X_{n+j}: tap pair kk of step j against the swapped Pair(n + j - 2 kk - 1) */
        const float gj[4] = {g0, g1, g2, g3};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const v2f gg = {gj[j], gj[j]};
#pragma unroll
          for (int kk = 0; kk < TPL / 2; kk++) {
            const v2f pp = P[(n + j - 2 * kk - 1) & 15];
            w2[kk] = __builtin_elementwise_fma(gg, __builtin_shufflevector(pp, pp, 1, 0), w2[kk]);
          }
        }
        ra = na; rb = nb; rc = nc; dq = nd;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  static __device__ __forceinline__ float4 out4(const float *out, const float *, int i) {
    return *reinterpret_cast<const float4 *>(out + i);
  }
};

