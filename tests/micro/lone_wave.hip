// Issue / latency model of ONE wave per SIMD on gfx950 -- the regime of the NLMS tail kernel
// (4096 channels = 1024 waves = one per SIMD).  Every mode runs an unrolled body of 64 instructions
// `iters` times and reports shader cycles per instruction (s_memtime) for 1 and 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tests/micro/lone_wave.hip -o tests/micro/lone_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int MODE>
__global__ void __launch_bounds__(64) k(float *out, long long *cyc, int iters, float a, float b) {
  float v[16];
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 0.001f + i;
  v2f p[8], q[8], r[8];
  for (int i = 0; i < 8; i++) {
    p[i] = v2f{v[2 * i], v[2 * i + 1]};
    q[i] = v2f{a + i, b - i};
    r[i] = v2f{b + i, a - i};
  }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // 16 independent plain fma x4
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(a), "v"(b));
    } else if (MODE == 1) {  // dependent plain fma
#pragma unroll
      for (int rr = 0; rr < 64; rr++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[0]) : "v"(a), "v"(b));
    } else if (MODE == 2) {  // 8 independent pk_fma, three distinct 64-bit sources, x8
#pragma unroll
      for (int rr = 0; rr < 8; rr++)
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(q[i]), "v"(r[(i + rr) & 7]));
    } else if (MODE == 3) {  // dependent pk_fma (accumulator chain, like the dot product)
#pragma unroll
      for (int rr = 0; rr < 64; rr++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[0]) : "v"(q[rr & 7]), "v"(r[(rr >> 3) & 7]));
    } else if (MODE == 4) {  // dependent DPP add with its two wait states
#pragma unroll
      for (int rr = 0; rr < 64; rr++) asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[0]));
    } else if (MODE == 5) {  // independent DPP adds (sources written long ago)
#pragma unroll
      for (int rr = 0; rr < 4; rr++)
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]) : "v"(v[(i + 8) & 15]));
    } else if (MODE == 6) {  // dependent pk_fma chain with one independent plain fma between links
#pragma unroll
      for (int rr = 0; rr < 32; rr++) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[0]) : "v"(q[rr & 7]), "v"(r[(rr >> 3) & 7]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[1 + (rr & 7)]) : "v"(a), "v"(b));
      }
    } else if (MODE == 7) {  // dependent DPP chain with two independent plain fma in the wait states
#pragma unroll
      for (int rr = 0; rr < 21; rr++) {
        asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[0]));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[1 + (rr & 3)]) : "v"(a), "v"(b));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[5 + (rr & 3)]) : "v"(a), "v"(b));
      }
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[9]) : "v"(a), "v"(b));
    } else if (MODE == 8) {  // two interleaved dependent pk_fma chains
#pragma unroll
      for (int rr = 0; rr < 32; rr++) {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[0]) : "v"(q[rr & 7]), "v"(r[(rr >> 3) & 7]));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[1]) : "v"(q[(rr + 3) & 7]), "v"(r[(rr >> 3) & 7]));
      }
    } else if (MODE == 9) {  // v_fmac_f32_dpp row_newbcast, dependent through the broadcast source
#pragma unroll
      for (int rr = 0; rr < 32; rr++) {
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[1]) : "v"(v[0]), "v"(a));
        asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(v[0]) : "v"(v[1]), "v"(v[2 + (rr & 7)]));
      }
    } else if (MODE == 10) {  // v_permlane16_swap, dependent
#pragma unroll
      for (int rr = 0; rr < 64; rr++) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[0]), "+v"(v[1]));
    } else if (MODE == 11) {  // plain fma reading an LDS result each: ds_read_b32 + use (latency of LDS in chain)
      extern __shared__ float sm[];
      sm[threadIdx.x] = v[0];
#pragma unroll
      for (int rr = 0; rr < 32; rr++) {
        float t;
        asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((int)(threadIdx.x * 4)));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[0]) : "v"(t), "v"(b));
      }
    } else if (MODE == 12) {  // the tail's block as it is: 6 dep pk + 5 dep dpp(+nops) + 2 dpp sub + 3 + 6 pk (3 chains of 2)
#pragma unroll
      for (int rr = 0; rr < 2; rr++) {
#pragma unroll
        for (int i = 0; i < 6; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[0]) : "v"(p[1 + (i >> 1)]), "v"(r[i]));
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n"
                     "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     : "=&v"(v[0]) : "v"(p[0][0]), "v"(p[0][1]));
        asm volatile("v_subrev_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(v[1]) : "v"(v[0]), "v"(a));
        asm volatile("v_subrev_f32_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf" : "=v"(v[2]) : "v"(v[0]), "v"(b));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[3]) : "v"(v[1]), "v"(a));
        asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(v[4]) : "v"(v[3]), "v"(b), "v"(v[2]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[5]) : "v"(v[4]), "v"(a));
        v2f g0 = v2f{v[3], v[3]}, g1 = v2f{v[5], v[5]};
#pragma unroll
        for (int i = 0; i < 3; i++) {
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[1 + i]) : "v"(g0), "v"(r[i]));
          asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[1 + i]) : "v"(g1), "v"(r[i + 3]));
        }
      }

    } else if (MODE == 13 || MODE == 14 || MODE == 15) {
      // lookahead form of the tail block as in rdsp_tail.hip (NlmsL): block A (RED + UPD + DOT), block B (REC);
      // 13: the two blocks alone; 14: plus the LDS traffic of an iteration (6 instructions); 15: 14 plus s_nop 0 x2
      extern __shared__ float sm[];
      const int la = (int)(threadIdx.x * 16);
      float t, dA0, dA1, u0, u1, e0, e1, g0, g1;
      v2f an = p[0], ac2, w0 = p[1], w1 = p[2], w2 = p[3], G2 = q[6], G1 = q[7];
      v2f ao = p[4];
      float4 ra = *(float4 *)&sm[threadIdx.x * 4], rb = ra;
      v2f dd = q[5], n0 = q[4], n1 = q[3];
#pragma unroll
      for (int rr = 0; rr < 2; rr++) {
        if (MODE >= 14) {
          asm volatile("ds_write_b64 %0, %1 offset:8192" ::"v"(la), "v"(dd));
          asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(ra) : "v"(la));
          asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(rb) : "v"(la));
          asm volatile("ds_read_b64 %0, %1 offset:3072" : "=v"(dd) : "v"(la));
          asm volatile("ds_read2_b32 %0, %1 offset0:4 offset1:5" : "=v"(n0) : "v"(la));
          asm volatile("ds_read_b64 %0, %1 offset:4096" : "=v"(n1) : "v"(la));
          asm volatile("s_waitcnt lgkmcnt(6)");
        }
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
            : [t] "=&v"(t), [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2), [an] "=&v"(an), [ac2] "=&v"(ac2)
            : [a0] "v"(ao[0]), [a1] "v"(ao[1]), [G2] "v"(G2), [G1] "v"(G1), [pm7] "v"(r[0]), [pm6] "v"(r[1]),
              [pm5] "v"(r[2]), [pm4] "v"(r[3]), [pm3] "v"(r[4]), [pm2] "v"(r[5]), [pm1] "v"(r[6]), [p0] "v"(r[7]),
              [pp1] "v"(n0), [pp2] "v"(n1));
        if (MODE == 15) asm volatile("s_nop 0");
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
              [g0o] "=&v"(g0), [g1o] "=&v"(g1), [ac2] "+v"(ac2), [an] "+v"(an)
            : [t] "v"(t), [d0] "v"(dd[0]), [d1] "v"(dd[1]), [w2] "v"(w2), [pm3] "v"(r[4]), [g2] "v"(G2[0]),
              [g1] "v"(G1[0]), [r2a] "v"(rb.x), [r3b] "v"(rb.z), [r1a] "v"(ra.z), [r2b] "v"(rb.y), [c0] "v"(ra.x),
              [r1b] "v"(ra.w), [c1] "v"(ra.y));
        if (MODE == 15) asm volatile("s_nop 0");
        { v2f ga, gb; ga[0] = g0; gb[0] = g1; G2 = ga; G1 = gb; }
        ao = an;
      }
      p[0] = an; p[1] = w0; p[2] = w1; p[3] = w2; v[14] += n0[0] + n1[1] + ra.x + rb.y + dd[0];
    } else if (MODE == 16 || MODE == 17) {
      // NlmsQ's four-step block as compiled (rdsp_tail.hip): DOT 12 pk, RED 8 DPP + nops, 4 bcast subs,
      // REC 10, UPD 12 pk.  17: the same plus 9 LDS instructions
      extern __shared__ float sm[];
      const int la = (int)(threadIdx.x * 16);
      v2f w0 = p[1], w1 = p[2], w2 = p[3], aA, aB;
      float ta, tb, u, d0, d1, d2, d3, g0, g1, g2, g3, e1, e2, e3, t2, t3;
      float4 ra = *(float4 *)&sm[threadIdx.x * 4], rb = ra, rc = ra, dq = ra;
      v2f n0 = q[4], n1 = q[3], n2 = q[2], n3 = q[1];
      for (int rr = 0; rr < 1; rr++) {
        if (MODE == 17) {
          asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(ra) : "v"(la));
          asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(rb) : "v"(la));
          asm volatile("ds_read_b128 %0, %1 offset:3072" : "=v"(rc) : "v"(la));
          asm volatile("ds_read_b128 %0, %1 offset:5120" : "=v"(dq) : "v"(la));
          asm volatile("ds_read2_b32 %0, %1 offset0:4 offset1:5" : "=v"(n0) : "v"(la));
          asm volatile("ds_read2_b32 %0, %1 offset0:6 offset1:7" : "=v"(n1) : "v"(la));
          asm volatile("ds_read2_b32 %0, %1 offset0:8 offset1:9" : "=v"(n2) : "v"(la));
          asm volatile("ds_read2_b32 %0, %1 offset0:10 offset1:11" : "=v"(n3) : "v"(la));
          asm volatile("ds_write_b64 %0, %1 offset:8192" ::"v"(la), "v"(n0));
          asm volatile("s_waitcnt lgkmcnt(8)");
        }
        asm volatile(
            "v_pk_mul_f32 %[aA], %[p0], %[w0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %[aB], %[p2], %[w0] op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %[aA], %[w0], %[p1], %[aA] op_sel:[1,0,0]\n\t"
            "v_pk_fma_f32 %[aB], %[w0], %[p3], %[aB] op_sel:[1,0,0]\n\t"
            "v_pk_fma_f32 %[aA], %[w1], %[p4], %[aA] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[aB], %[w1], %[p0], %[aB] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[aA], %[w1], %[p5], %[aA] op_sel:[1,0,0]\n\t"
            "v_pk_fma_f32 %[aB], %[w1], %[p1], %[aB] op_sel:[1,0,0]\n\t"
            "v_pk_fma_f32 %[aA], %[w2], %[p6], %[aA] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[aB], %[w2], %[p4], %[aB] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[aA], %[w2], %[p7], %[aA] op_sel:[1,0,0]\n\t"
            "v_pk_fma_f32 %[aB], %[w2], %[p5], %[aB] op_sel:[1,0,0]"
            : [aA] "=&v"(aA), [aB] "=&v"(aB)
            : [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [p0] "v"(r[0]), [p1] "v"(r[1]), [p2] "v"(r[2]), [p3] "v"(r[3]),
              [p4] "v"(r[4]), [p5] "v"(r[5]), [p6] "v"(r[6]), [p7] "v"(r[7]));
        asm volatile(
            "s_nop 1\n\t"
            "v_add_f32_dpp %[ta], %[a0], %[a0] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %[tb], %[b0], %[b0] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %[ta], %[a1], %[a1] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_f32_dpp %[tb], %[b1], %[b1] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[u], %[ta], %[ta] row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %[u], %[tb], %[tb] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[u], %[u], %[u] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_add_f32_dpp %[u], %[u], %[u] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\t"
            "v_subrev_f32_dpp %[d0], %[u], %[x0] row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
            "v_subrev_f32_dpp %[d1], %[u], %[x1] row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_mul_f32 %[g0], %[c0], %[d0]\n\t"
            "v_subrev_f32_dpp %[d2], %[u], %[x2] row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_fma_f32 %[e1], -%[g0], %[r11], %[d1]\n\t"
            "v_subrev_f32_dpp %[d3], %[u], %[x3] row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
            "v_fma_f32 %[t2], -%[g0], %[r22], %[d2]\n\t"
            "v_mul_f32 %[g1], %[c1], %[e1]\n\t"
            "v_fma_f32 %[t3], -%[g0], %[r33], %[d3]\n\t"
            "v_fma_f32 %[e2], -%[g1], %[r12], %[t2]\n\t"
            "v_fma_f32 %[t3], -%[g1], %[r23], %[t3]\n\t"
            "v_mul_f32 %[g2], %[c2], %[e2]\n\t"
            "v_fma_f32 %[e3], -%[g2], %[r13], %[t3]\n\t"
            "v_mul_f32 %[g3], %[c3], %[e3]"
            : [ta] "=&v"(ta), [tb] "=&v"(tb), [u] "=&v"(u), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3),
              [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3), [e1] "=&v"(e1), [e2] "=&v"(e2), [e3] "=&v"(e3),
              [t2] "=&v"(t2), [t3] "=&v"(t3)
            : [a0] "v"(aA[0]), [a1] "v"(aA[1]), [b0] "v"(aB[0]), [b1] "v"(aB[1]), [x0] "v"(dq.x), [x1] "v"(dq.y), [x2] "v"(dq.z),
              [x3] "v"(dq.w), [c0] "v"(ra.x), [c1] "v"(ra.y), [c2] "v"(ra.z), [c3] "v"(ra.w), [r11] "v"(rb.x), [r12] "v"(rb.y),
              [r13] "v"(rb.z), [r22] "v"(rb.w), [r23] "v"(rc.x), [r33] "v"(rc.y));
        {
          v2f G0, G1, G2, G3; G0[0] = g0; G1[0] = g1; G2[0] = g2; G3[0] = g3;
#define U " op_sel:[0,1,0] op_sel_hi:[0,0,1]\n\t"
          asm volatile(
              "v_pk_fma_f32 %[w0], %[G0], %[p1], %[w0]" U "v_pk_fma_f32 %[w1], %[G0], %[p5], %[w1]" U "v_pk_fma_f32 %[w2], %[G0], %[p7], %[w2]" U
              "v_pk_fma_f32 %[w0], %[G1], %[p0], %[w0]" U "v_pk_fma_f32 %[w1], %[G1], %[p4], %[w1]" U "v_pk_fma_f32 %[w2], %[G1], %[p6], %[w2]" U
              "v_pk_fma_f32 %[w0], %[G2], %[p3], %[w0]" U "v_pk_fma_f32 %[w1], %[G2], %[p1], %[w1]" U "v_pk_fma_f32 %[w2], %[G2], %[p5], %[w2]" U
              "v_pk_fma_f32 %[w0], %[G3], %[p2], %[w0]" U "v_pk_fma_f32 %[w1], %[G3], %[p0], %[w1]" U "v_pk_fma_f32 %[w2], %[G3], %[p4], %[w2]"
              : [w0] "+v"(w0), [w1] "+v"(w1), [w2] "+v"(w2)
              : [G0] "v"(G0), [G1] "v"(G1), [G2] "v"(G2), [G3] "v"(G3), [p0] "v"(r[0]), [p1] "v"(r[1]), [p2] "v"(r[2]), [p3] "v"(r[3]),
                [p4] "v"(r[4]), [p5] "v"(r[5]), [p6] "v"(r[6]), [p7] "v"(r[7]));
#undef U
        }
      }
      p[1] = w0; p[2] = w1; p[3] = w2; v[14] += n0[0] + n1[1] + n2[0] + n3[1] + e1 + e2 + e3 + d0;
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; i++) s += v[i];
  for (int i = 0; i < 8; i++) s += p[i][0] + p[i][1];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int instr_per_iter) {
  float *d;
  long long *c;
  hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
  hipMalloc(&c, 256 * 4 * 8 * 8);
  for (int wps : {1, 2, 3}) {
    const int grid = 256 * 4 * wps, iters = 4000;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 256, 0, d, c, 100, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 256, 0, d, c, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), c, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (long long x : h) avg += (double)x;
    avg /= grid;
    printf("%-52s waves/SIMD %d: %.3f ms  %6.2f memtime ticks/instr  %6.2f ns/instr\n", name, wps, ms,
           avg / ((double)iters * instr_per_iter), ms * 1e6 / ((double)iters * instr_per_iter));
  }
  hipFree(d); hipFree(c);
}
int main() {
  run<0>("v_fma_f32 independent", 64);
  run<1>("v_fma_f32 dependent", 64);
  run<2>("v_pk_fma_f32 independent, 3 distinct sources", 64);
  run<3>("v_pk_fma_f32 dependent (accumulator chain)", 64);
  run<4>("s_nop 1 + v_add_f32_dpp dependent (per pair)", 64);
  run<5>("v_add_f32_dpp independent", 64);
  run<6>("dep pk_fma + 1 indep fma (per pair)", 32);
  run<7>("dep dpp + 2 indep fma (per triple)", 21);
  run<8>("two interleaved dep pk_fma chains (per pair)", 32);
  run<9>("mul + nop + v_fmac_dpp newbcast dependent (per pair)", 32);
  run<10>("v_permlane16_swap dependent", 64);
  run<11>("ds_read_b32 + wait + fma dependent (per pair)", 32);
  run<12>("tail block as is (per 2-step block)", 2);
  run<13>("NlmsL blocks A+B (per 2-step iteration)", 2);
  run<14>("NlmsL blocks + 6 LDS instructions", 2);
  run<15>("NlmsL blocks + LDS + 2 s_nop", 2);
  run<16>("NlmsQ four-step block as compiled (per block)", 1);
  run<17>("NlmsQ block + 9 LDS instructions", 1);
  return 0;
}
