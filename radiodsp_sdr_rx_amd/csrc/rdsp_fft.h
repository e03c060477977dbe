/*
 * rdsp_fft.h -- workgroup-cooperative complex FFT for the overlap-save filter
 * (replaces the arm_cfft_f32 calls at RDSP_convolutional.h:291,309 and
 * backup/RDSP_convolutional_spec.h:179,243).
 *
 * Design (MI355X): one N-point transform is shared by NT = N/P threads, each
 * holding P points in registers.  Forward = in-place decimation-in-frequency
 * passes of radix P (last pass radix N / P^(k-1)), which leaves bin k at the
 * digit-reversed position; the filter mask is stored in that order, so no
 * reordering pass exists.  Inverse = the exact transpose (decimation-in-time)
 * and lands in natural time order.  Between passes the data goes through LDS
 * under one address map per exchange (FftPlan::xshift / xmul): bank-conflict-free
 * for the one-wave plans, phi(i) = i + i/P for the four-wave ones.  Per-thread
 * twiddles live in registers for the whole kernel.
 *
 * Everything here is __host__ __device__ so tests/host_fft_check.cpp can run
 * the same code thread by thread on the CPU.
 */
#ifndef RDSP_FFT_H
#define RDSP_FFT_H

#include <hip/hip_runtime.h>
#include <math.h>

#define RDSP_HD __host__ __device__ __forceinline__

namespace rdsp {

RDSP_HD float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
RDSP_HD float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
/* Complex products as two packed instructions on the device:
 *     t = (a.y b.y, a.y b.x)            v_pk_mul_f32, op_sel picks the halves
 *     r = (a.x b.x -/+ t.lo, +/- a.x b.y + t.hi)   v_pk_fma_f32 with neg modifiers
 * The compiler's own lowering of the scalar formula is pk_mul + two half-used pk_fma + a
 * v_mov to merge them (4 instructions; ~80 complex products per chunk and lane). */
typedef float rdsp_v2f __attribute__((ext_vector_type(2)));
RDSP_HD float2 cmul(float2 a, float2 b) {
#ifdef __HIP_DEVICE_COMPILE__
  const rdsp_v2f av = {a.x, a.y}, bv = {b.x, b.y};
  rdsp_v2f r;
  /* one statement for the dependent pair, the product going through the result register: the hazard
   * recognizer cannot see inside an asm statement and puts a wait state between two of them that
   * touch the same register (a result read, or a scratch register written again); the hardware
   * needs none between packed VALU operations (278 such s_nop per frame loop with two statements
   * per product) */
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
      : "=&v"(r) : "v"(av), "v"(bv));
  return make_float2(r.x, r.y);
#else
  return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
#endif
}
/* cmul with a wave-uniform second factor kept in scalar registers (one SGPR pair per packed
 * instruction is within the constant-bus limit); same arithmetic as cmul() */
RDSP_HD float2 cmul_uniform(float2 a, float2 b) {
#ifdef __HIP_DEVICE_COMPILE__
  const rdsp_v2f av = {a.x, a.y}, bv = {b.x, b.y};
  rdsp_v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[0,0,1]"
      : "=&v"(r) : "v"(av), "s"(bv));
  return make_float2(r.x, r.y);
#else
  return cmul(a, b);
#endif
}
/* acc + a * b: two packed FMAs */
RDSP_HD float2 cmac(float2 acc, float2 a, float2 b) {
#ifdef __HIP_DEVICE_COMPILE__
  const rdsp_v2f av = {a.x, a.y}, bv = {b.x, b.y};
  rdsp_v2f r = {acc.x, acc.y};
  /* r = (acc.x - a.y b.y, acc.y + a.y b.x);  r = (a.x b.x + r.x, a.x b.y + r.y) */
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]"
      : "+v"(r) : "v"(av), "v"(bv));
  return make_float2(r.x, r.y);
#else
  return make_float2(fmaf(a.x, b.x, fmaf(-a.y, b.y, acc.x)), fmaf(a.x, b.y, fmaf(a.y, b.x, acc.y)));
#endif
}
/* a * conj(b) with a wave-uniform b in scalar registers */
RDSP_HD float2 cmulc_uniform(float2 a, float2 b) {
#ifdef __HIP_DEVICE_COMPILE__
  const rdsp_v2f av = {a.x, a.y}, bv = {b.x, b.y};
  rdsp_v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
      : "=&v"(r) : "v"(av), "s"(bv));
  return make_float2(r.x, r.y);
#else
  return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(-a.x, b.y, a.y * b.x));
#endif
}
RDSP_HD float2 cmulc(float2 a, float2 b) { /* a * conj(b) */
#ifdef __HIP_DEVICE_COMPILE__
  const rdsp_v2f av = {a.x, a.y}, bv = {b.x, b.y};
  rdsp_v2f r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
      : "=&v"(r) : "v"(av), "v"(bv));
  return make_float2(r.x, r.y);
#else
  return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(-a.x, b.y, a.y * b.x));
#endif
}

/* multiply by w16^M (forward, w = exp(-2*pi*i/16)) or its conjugate (INV) */
template <int M, bool INV>
RDSP_HD float2 mul_w16(float2 a) {
  constexpr float C1 = 0.92387953251128674f; /* cos(pi/8) */
  constexpr float S1 = 0.38268343236508977f; /* sin(pi/8) */
  constexpr float C2 = 0.70710678118654752f; /* cos(pi/4) */
  constexpr int m = M & 15;
  if constexpr (m == 0) return a;
  else if constexpr (m == 4) return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
  else if constexpr (m == 8) return make_float2(-a.x, -a.y);
  else if constexpr (m == 12) return INV ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
  else if constexpr (m == 2)
    return INV ? make_float2(C2 * (a.x - a.y), C2 * (a.x + a.y))
               : make_float2(C2 * (a.x + a.y), C2 * (a.y - a.x));
  else if constexpr (m == 6)
    return INV ? make_float2(-C2 * (a.x + a.y), C2 * (a.x - a.y))
               : make_float2(C2 * (a.y - a.x), -C2 * (a.x + a.y));
  else {
    /* w16^m = cos(2 pi m/16) - i sin(2 pi m/16) */
    constexpr float cr = (m == 1) ? C1 : (m == 3) ? S1 : (m == 5) ? -S1 : (m == 7) ? -C1
                       : (m == 9) ? -C1 : (m == 11) ? -S1 : (m == 13) ? S1 : C1;
    constexpr float sr = (m == 1) ? S1 : (m == 3) ? C1 : (m == 5) ? C1 : (m == 7) ? S1
                       : (m == 9) ? -S1 : (m == 11) ? -C1 : (m == 13) ? -C1 : -S1;
    /* forward: (cr - i sr); inverse: (cr + i sr) */
    return INV ? make_float2(a.x * cr - a.y * sr, a.x * sr + a.y * cr)
               : make_float2(a.x * cr + a.y * sr, a.y * cr - a.x * sr);
  }
}

/* One LDS element as ONE ds_read_b64.  Left alone, the compiler pairs the passes' reads into
 * ds_read2_b64, which the LDS serves at 128 B/clk (8 cycles per wave-instruction) where two
 * ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS table) -- and the front kernels keep the LDS
 * array of a CU busy 65-70 % of the time (PMC, round 3).  A volatile access is never merged. */
RDSP_HD float2 lds_ld(const float2 *p) {
#ifdef __HIP_DEVICE_COMPILE__
  typedef const volatile __attribute__((address_space(3))) rdsp_v2f *LP;
  const rdsp_v2f v = *(LP)p;
  return make_float2(v.x, v.y);
#else
  return *p;
#endif
}

template <bool INV>
RDSP_HD void dft2(float2 &a, float2 &b) {
  float2 t = a;
  a = cadd(t, b);
  b = csub(t, b);
}

#ifdef __HIP_DEVICE_COMPILE__
/* Device butterflies on native <2 x float> values: every complex add is one
 * v_pk_add_f32, and the rotations by -i/+i and by w8^1, w8^3 are folded into the
 * op_sel / neg modifiers of the add that consumes them, written out by hand --
 * left to the SLP vectoriser the scalar formulas pair real parts of different
 * values and glue the halves back together with v_mov (59 of the 260 instructions
 * of a 512-point forward transform). */
namespace pk {
typedef rdsp_v2f V;
RDSP_HD V ld(float2 a) { return V{a.x, a.y}; }
RDSP_HD float2 st(V a) { return make_float2(a.x, a.y); }
/* a + r*b with r = -i (forward) or +i (INV); add_rot<!INV> is a - r*b */
template <bool INV>
RDSP_HD V add_rot(V a, V b) {
  V r;
  if constexpr (INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
/* sqrt(2) * a * w8^1: (x + y, y - x) forward, (x - y, x + y) inverse */
template <bool INV>
RDSP_HD V rot8_1(V a) { return add_rot<INV>(a, a); }
/* sqrt(2) * a * w8^3: (y - x, -x - y) forward, (-x - y, x - y) inverse */
template <bool INV>
RDSP_HD V rot8_3(V a) {
  V r;
  if constexpr (INV) asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(r) : "v"(a));
  else asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,1]" : "=v"(r) : "v"(a));
  return r;
}
template <bool INV>
RDSP_HD void dft4(V &a0, V &a1, V &a2, V &a3) {
  const V t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, t3 = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = add_rot<INV>(t1, t3);
  a3 = add_rot<!INV>(t1, t3);
}
} // namespace pk
#endif

/* 4-point DFT, natural order in and out, on v[0], v[ST], v[2ST], v[3ST] */
template <bool INV, int ST>
RDSP_HD void dft4(float2 *v) {
#ifdef __HIP_DEVICE_COMPILE__
  pk::V a0 = pk::ld(v[0]), a1 = pk::ld(v[ST]), a2 = pk::ld(v[2 * ST]), a3 = pk::ld(v[3 * ST]);
  pk::dft4<INV>(a0, a1, a2, a3);
  v[0] = pk::st(a0);
  v[ST] = pk::st(a1);
  v[2 * ST] = pk::st(a2);
  v[3 * ST] = pk::st(a3);
#else
  float2 t0 = cadd(v[0], v[2 * ST]);
  float2 t1 = csub(v[0], v[2 * ST]);
  float2 t2 = cadd(v[ST], v[3 * ST]);
  float2 t3 = csub(v[ST], v[3 * ST]);
  float2 t3r = mul_w16<4, INV>(t3); /* -i*t3 (fwd) / +i*t3 (inv) */
  v[0] = cadd(t0, t2);
  v[2 * ST] = csub(t0, t2);
  v[ST] = cadd(t1, t3r);
  v[3 * ST] = csub(t1, t3r);
#endif
}

template <int R, bool INV>
struct Dft;

template <bool INV>
struct Dft<2, INV> {
  static RDSP_HD void run(float2 *v) { dft2<INV>(v[0], v[1]); }
};

template <bool INV>
struct Dft<4, INV> {
  static RDSP_HD void run(float2 *v) { dft4<INV, 1>(v); }
};

template <bool INV>
struct Dft<8, INV> {
  /* n = n1 + 2*n2, k = 4*k1 + k2 */
  static RDSP_HD void run(float2 *v) {
#ifdef __HIP_DEVICE_COMPILE__
    using pk::V;
    V x0 = pk::ld(v[0]), x1 = pk::ld(v[1]), x2 = pk::ld(v[2]), x3 = pk::ld(v[3]);
    V x4 = pk::ld(v[4]), x5 = pk::ld(v[5]), x6 = pk::ld(v[6]), x7 = pk::ld(v[7]);
    pk::dft4<INV>(x0, x2, x4, x6); /* A[k2] at x(2 k2)     */
    pk::dft4<INV>(x1, x3, x5, x7); /* B[k2] at x(2 k2 + 1) */
    const V b1 = pk::rot8_1<INV>(x3), b3 = pk::rot8_3<INV>(x7);
    const V c2 = {0.70710678118654752f, 0.70710678118654752f};
    v[0] = pk::st(x0 + x1);
    v[4] = pk::st(x0 - x1);
    v[1] = pk::st(x2 + b1 * c2);
    v[5] = pk::st(x2 - b1 * c2);
    v[2] = pk::st(pk::add_rot<INV>(x4, x5));
    v[6] = pk::st(pk::add_rot<!INV>(x4, x5));
    v[3] = pk::st(x6 + b3 * c2);
    v[7] = pk::st(x6 - b3 * c2);
#else
    dft4<INV, 2>(v);     /* n1 = 0: elements 0,2,4,6 -> A[0][k2] at 2*k2   */
    dft4<INV, 2>(v + 1); /* n1 = 1: elements 1,3,5,7 -> A[1][k2] at 1+2*k2 */
    v[3] = mul_w16<2, INV>(v[3]); /* w8^1 */
    v[5] = mul_w16<4, INV>(v[5]); /* w8^2 */
    v[7] = mul_w16<6, INV>(v[7]); /* w8^3 */
    float2 o[8];
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) {
      o[k2] = cadd(v[2 * k2], v[2 * k2 + 1]);
      o[4 + k2] = csub(v[2 * k2], v[2 * k2 + 1]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = o[i];
#endif
  }
};

template <bool INV>
struct Dft<16, INV> {
  /* n = n1 + 4*n2, k = 4*k1 + k2 */
  static RDSP_HD void run(float2 *v) {
#ifdef __HIP_DEVICE_COMPILE__
    /* On native <2 x float> values like Dft<8>: 77 packed instructions where the scalar formulas below
     * compile to 100 (the +-i rotations and the 1/sqrt(2) of w16^2, w16^4, w16^6 ride on the adds that
     * consume them; w16^1, w16^3, w16^9 are two packed instructions each against scalar-register constants) */
    using pk::V;
    constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, C2 = 0.70710678118654752f;
    const V c2 = {C2, C2};
    V x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = pk::ld(v[i]);
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) pk::dft4<INV>(x[n1], x[n1 + 4], x[n1 + 8], x[n1 + 12]); /* A[n1][k2] at x[n1 + 4 k2] */
    auto tw = [](V a, float cr, float sr) { /* a * (cr - i sr), conjugated for the inverse */
      return pk::ld(cmul_uniform(pk::st(a), make_float2(cr, INV ? sr : -sr)));
    };
    auto fin = [&](int k2, V t0, V t1, V t2, V t3) { /* X[4 k1 + k2] = second-stage output k1 */
      v[k2] = pk::st(t0 + t2);
      v[8 + k2] = pk::st(t0 - t2);
      v[4 + k2] = pk::st(pk::add_rot<INV>(t1, t3));
      v[12 + k2] = pk::st(pk::add_rot<!INV>(t1, t3));
    };
    fin(0, x[0] + x[2], x[0] - x[2], x[1] + x[3], x[1] - x[3]);
    { /* k2 = 1: twiddles w^0, w^1, w^2, w^3 */
      const V a1 = tw(x[5], C1, S1), a3 = tw(x[7], S1, C1), r2 = pk::rot8_1<INV>(x[6]); /* x6 w^2 = c2 r2 */
      fin(1, x[4] + r2 * c2, x[4] - r2 * c2, a1 + a3, a1 - a3);
    }
    { /* k2 = 2: w^0, w^2, w^4, w^6 */
      const V r9 = pk::rot8_1<INV>(x[9]), r11 = pk::rot8_3<INV>(x[11]);
      const V t0 = pk::add_rot<INV>(x[8], x[10]), t1 = pk::add_rot<!INV>(x[8], x[10]); /* x10 w^4 = -+i x10 */
      fin(2, t0, t1, (r9 + r11) * c2, (r9 - r11) * c2);
    }
    { /* k2 = 3: w^0, w^3, w^6, w^9 */
      const V a1 = tw(x[13], S1, C1), a3 = tw(x[15], -C1, -S1), r2 = pk::rot8_3<INV>(x[14]); /* x14 w^6 = c2 r2 */
      fin(3, x[12] + r2 * c2, x[12] - r2 * c2, a1 + a3, a1 - a3);
    }
    return;
#endif
    dft4<INV, 4>(v);
    dft4<INV, 4>(v + 1);
    dft4<INV, 4>(v + 2);
    dft4<INV, 4>(v + 3);
    /* element n1 + 4*k2 holds A[n1][k2]; multiply by w16^(n1*k2) */
    v[5] = mul_w16<1, INV>(v[5]);
    v[9] = mul_w16<2, INV>(v[9]);
    v[13] = mul_w16<3, INV>(v[13]);
    v[6] = mul_w16<2, INV>(v[6]);
    v[10] = mul_w16<4, INV>(v[10]);
    v[14] = mul_w16<6, INV>(v[14]);
    v[7] = mul_w16<3, INV>(v[7]);
    v[11] = mul_w16<6, INV>(v[11]);
    v[15] = mul_w16<9, INV>(v[15]);
    dft4<INV, 1>(v);      /* k2 = 0: elements 0..3  -> k1 at k1      */
    dft4<INV, 1>(v + 4);  /* k2 = 1: elements 4..7                   */
    dft4<INV, 1>(v + 8);
    dft4<INV, 1>(v + 12);
    float2 o[16];
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++)
#pragma unroll
      for (int k2 = 0; k2 < 4; k2++) o[4 * k1 + k2] = v[4 * k2 + k1];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = o[i];
  }
};

/* ---- plan constants ----------------------------------------------------- */
constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }

template <int N, int P>
struct FftPlan {
  static constexpr int NT = N / P;
  static constexpr int LOGP = ilog2(P);
  static constexpr int LOGN = ilog2(N);
  static constexpr int NFULL = LOGN / LOGP;                 /* radix-P passes */
  static constexpr int RL = (LOGN % LOGP) ? (1 << (LOGN % LOGP)) : P; /* last radix */
  static constexpr int NP = (LOGN % LOGP) ? NFULL + 1 : NFULL;        /* passes */
  static constexpr int NTW = NP - 1;                        /* twiddled passes */
  static constexpr int WB = N + N / P;                      /* padded LDS entries */
  /* span of pass p (p < NP-1): N / P^(p+1); last pass: 1 */
  static constexpr int span(int p) { return p >= NP - 1 ? 1 : (N >> (LOGP * (p + 1))); }
  /* LDS map of exchange x (what pass x writes and pass x+1 reads; the inverse the other way round):
   *     A_x(i) = i + xmul(x) * (i >> xshift(x)).
   * One map for all exchanges, phi(i) = i + i/P, leaves the stride-1 lane pattern of pass 0 two-way
   * conflicting whatever the padding (exhaustive over two- and three-term additive maps); but an exchange
   * rewrites the whole buffer, so each may have a map of its own, and with these every access of the
   * one-wave plans is conflict-free under the bank rules of the instructions the passes use
   * (ds_write_b64: 16 contiguous lanes over 32 banks; ds_read_b64: 32 lanes over 64 banks --
   * MI355X_MICROARCH.md, LDS table; found by tests/micro/lds_model.py).  All of them fit in WB.
   * The four-wave plans keep phi throughout: a pass that reads under one map and writes under another
   * is not in place, and between waves that would take a second barrier per pass. */
  static constexpr bool PERX = (NT == 64);
  static constexpr int xshift(int x) {
    if (!PERX) return LOGP;
    if (N == 256 && P == 4) return x == 0 ? 6 : x == 1 ? 4 : 2;
    if (N == 512 && P == 8) return x == 0 ? 6 : 3;
    if (N == 1024 && P == 16) return x == 0 ? 6 : 4;
    return LOGP;
  }
  static constexpr int xmul(int x) {
    if (!PERX) return 1;
    if (N == 256 && P == 4) return x == 0 ? 16 : x == 1 ? 4 : 1;
    if (N == 512 && P == 8) return x == 0 ? 8 : 1;
    if (N == 1024 && P == 16) return x == 0 ? 4 : 1;
    return 1;
  }
};

template <int P>
RDSP_HD int phi(int i) { return i + (i >> ilog2(P)); }

/* natural bin index held at position i after the forward transform */
template <int N, int P>
RDSP_HD int bin_of_pos(int i) {
  using PL = FftPlan<N, P>;
  int k = 0, mult = 1;
#pragma unroll
  for (int p = 0; p < PL::NP; p++) {
    const int s = PL::span(p);
    const int R = (p == PL::NP - 1) ? PL::RL : P;
    int digit = (i / s) % R;
    k += digit * mult;
    mult *= R;
  }
  return k;
}

/* twiddles of thread t: tw[p][k-1] = exp(-2*pi*i * o*k / (P*s_p)), o = t % s_p */
template <int N, int P>
RDSP_HD void make_twiddles(int t, float2 (*tw)[P - 1]) {
  using PL = FftPlan<N, P>;
#pragma unroll
  for (int p = 0; p < PL::NTW; p++) {
    const int s = PL::span(p);
    const int S = P * s;
    int o = t % s;
#pragma unroll
    for (int k = 1; k < P; k++) {
      int m = (o * k) % S;
#ifdef __HIP_DEVICE_COMPILE__
      float sn, cs;
      sincospif(-2.0f * (float)m / (float)S, &sn, &cs);
      tw[p][k - 1] = make_float2(cs, sn);
#else
      double a = -2.0 * 3.14159265358979323846 * (double)m / (double)S;
      tw[p][k - 1] = make_float2((float)cos(a), (float)sin(a));
#endif
    }
  }
}

/* Twiddles of one pass from their base w1 = w^o by a short product chain
 * (w2 = w1^2, w3 = w2 w1, w4 = w2^2, w5 = w4 w1, w6 = w3^2, w7 = w4 w3, ...): at most
 * 4 roundings deep for k <= 15, ~2e-7.  Keeps one float2 per pass in registers
 * instead of P-1 (the kernel trades ~6 complex multiplies per pass for 24 VGPRs). */
template <int P>
RDSP_HD void twiddle_chain(float2 w1, float2 *tw /* [P-1]: w^1 .. w^(P-1) */) {
  tw[0] = w1;
  if constexpr (P >= 4) {
    tw[1] = cmul(w1, w1);
    tw[2] = cmul(tw[1], w1);
  }
  if constexpr (P >= 8) {
    tw[3] = cmul(tw[1], tw[1]);
    tw[4] = cmul(tw[3], w1);
    tw[5] = cmul(tw[2], tw[2]);
    tw[6] = cmul(tw[3], tw[2]);
  }
  if constexpr (P >= 16) {
    tw[7] = cmul(tw[3], tw[3]);
#pragma unroll
    for (int k = 9; k <= 15; k++) tw[k - 1] = cmul(tw[7], tw[k - 9]);
  }
}

/* base twiddle of every twiddled pass: w1[p] = exp(-2*pi*i * o / (P*s_p)) */
template <int N, int P>
RDSP_HD void make_twiddle_bases(int t, float2 *w1) {
  using PL = FftPlan<N, P>;
#pragma unroll
  for (int p = 0; p < PL::NTW; p++) {
    const int s = PL::span(p);
    const int S = P * s;
    int o = t % s;
#ifdef __HIP_DEVICE_COMPILE__
    float sn, cs;
    sincospif(-2.0f * (float)o / (float)S, &sn, &cs);
    w1[p] = make_float2(cs, sn);
#else
    double a = -2.0 * 3.14159265358979323846 * (double)o / (double)S;
    w1[p] = make_float2((float)cos(a), (float)sin(a));
#endif
  }
}

/* LDS addressing.  A thread touches, in pass p, the positions base_p + j*s_p
 * (j = 0..P-1); through a padded map A(i) = i + c (i >> a) that is
 *     A(base_p) + j*s_p + c ((j*s_p) >> a)   (base_p has no bits in j's field and
 * s_p is a power of two, so nothing carries across the shift), i.e. a per-thread
 * constant that is computed once per kernel plus a compile-time offset the
 * assembler folds into the ds_read/ds_write immediate.
 *
 * ALIAS = true places the work buffer inside the polyphase planes of the FIR
 * (rdsp_front.h): after the FIR of a chunk only the first 17 entries of each of
 * the 8 planes (the history) are live, so the N/8 elements [p*N/8, (p+1)*N/8) go
 * to plane p behind its history.  Offsets stay "per-thread constant + compile-time
 * constant" because N/8 is a multiple of every group size that does not span it. */
#ifndef RDSP_XP
#define RDSP_XP 84
#endif
template <int N, int P, bool ALIAS>
struct WbMap {
  using PL = FftPlan<N, P>;
  static constexpr int SEG = N / 8;              /* elements per plane          */
  static constexpr int CS = SEG + SEG / P;       /* padded float2 per plane     */
  static constexpr int PLANE = 2 * RDSP_XP;      /* float2 per plane            */
  static constexpr int HIST = 2 * 17;            /* float2 of history per plane */
  static_assert(!ALIAS || CS <= PLANE - HIST, "work buffer segment fits behind the history");
  /* X: the exchange (ALIAS keeps phi for every exchange: its planes are cut for it) */
  template <int X>
  static RDSP_HD int base(int elem) {
    if constexpr (ALIAS) return phi<P>(elem) + (elem / SEG) * (PLANE - CS) + HIST;
    else return elem + PL::xmul(X) * (elem >> PL::xshift(X));
  }
  template <int X>
  static constexpr int off(int j, int s) {
    return ALIAS ? j * s + (j * s) / P + ((j * s) / SEG) * (PLANE - CS)
                 : j * s + PL::xmul(X) * ((j * s) >> PL::xshift(X));
  }
};

template <int N, int P, bool ALIAS = false>
struct LdsBases {
  /* mapped base of every pass on its input side (exchange p-1; last pass: element t*P) and on its
   * output side (exchange p); equal, and one register, wherever the two exchanges share a map */
  int bi[FftPlan<N, P>::NP];
  int bo[FftPlan<N, P>::NP];
};

template <int N, int P, bool A, int PIDX = 0>
RDSP_HD void make_lds_bases(int t, LdsBases<N, P, A> &lb) {
  using PL = FftPlan<N, P>;
  if constexpr (PIDX < PL::NP - 1) {
    constexpr int s = PL::span(PIDX);
    const int base = (t / s) * P * s + (t % s);
    lb.bi[PIDX] = WbMap<N, P, A>::template base<(PIDX > 0 ? PIDX - 1 : 0)>(base);
    lb.bo[PIDX] = WbMap<N, P, A>::template base<PIDX>(base);
    make_lds_bases<N, P, A, PIDX + 1>(t, lb);
  } else {
    static_assert(PL::xshift(PL::NP - 2) >= PL::LOGP || A, "a thread's P points of the last pass are contiguous");
    lb.bi[PL::NP - 1] = lb.bo[PL::NP - 1] = WbMap<N, P, A>::template base<PL::NP - 2>(t * P);
  }
}

/* ---- forward ------------------------------------------------------------- */
/* pass 0: v[] already loaded with x[t + j*NT] (j = 0..P-1) */
template <int N, int P, bool A>
RDSP_HD void fwd_pass0_store(const LdsBases<N, P, A> &lb, float2 *v, float2 *wb, const float2 *twp) {
  using PL = FftPlan<N, P>;
  Dft<P, false>::run(v);
#pragma unroll
  for (int k = 1; k < P; k++) v[k] = cmul(v[k], twp[k - 1]);
#pragma unroll
  for (int j = 0; j < P; j++) wb[lb.bo[0] + WbMap<N, P, A>::template off<0>(j, PL::span(0))] = v[j];
}

/* middle pass p (1 <= p <= NP-2): reads under the map of exchange p-1, writes under that of exchange p
 * (in place where the two are one map; otherwise every thread of the transform has read before any
 * writes -- one wave, whose LDS operations execute in order) */
template <int N, int P, int PIDX, bool A>
RDSP_HD void fwd_mid_load(const LdsBases<N, P, A> &lb, float2 *v, const float2 *wb) {
  constexpr int s = FftPlan<N, P>::span(PIDX);
#pragma unroll
  for (int j = 0; j < P; j++) v[j] = lds_ld(&wb[lb.bi[PIDX] + WbMap<N, P, A>::template off<PIDX - 1>(j, s)]);
}
template <int N, int P, int PIDX, bool A>
RDSP_HD void fwd_mid_store(const LdsBases<N, P, A> &lb, float2 *v, float2 *wb, const float2 *twp) {
  constexpr int s = FftPlan<N, P>::span(PIDX);
  Dft<P, false>::run(v);
#pragma unroll
  for (int k = 1; k < P; k++) v[k] = cmul(v[k], twp[k - 1]);
#pragma unroll
  for (int j = 0; j < P; j++) wb[lb.bo[PIDX] + WbMap<N, P, A>::template off<PIDX>(j, s)] = v[j];
}
template <int N, int P, int PIDX, bool A>
RDSP_HD void fwd_pass_mid(const LdsBases<N, P, A> &lb, float2 *wb, const float2 *twp) {
  float2 v[P];
  fwd_mid_load<N, P, PIDX, A>(lb, v, wb);
  fwd_mid_store<N, P, PIDX, A>(lb, v, wb, twp);
}

/* last pass: loads positions t*P .. t*P+P-1, leaves the spectrum in v[] */
template <int N, int P, bool A>
RDSP_HD void fwd_pass_last(const LdsBases<N, P, A> &lb, float2 *v, const float2 *wb) {
  using PL = FftPlan<N, P>;
#pragma unroll
  for (int e = 0; e < P; e++) v[e] = lds_ld(&wb[lb.bi[PL::NP - 1] + e]);
#pragma unroll
  for (int q = 0; q < P / PL::RL; q++) Dft<PL::RL, false>::run(v + q * PL::RL);
}

/* ---- inverse (unnormalised; 1/N is folded into the mask) ------------------ */
template <int N, int P, bool A>
RDSP_HD void inv_pass_last(const LdsBases<N, P, A> &lb, float2 *v, float2 *wb) {
  using PL = FftPlan<N, P>;
#pragma unroll
  for (int q = 0; q < P / PL::RL; q++) Dft<PL::RL, true>::run(v + q * PL::RL);
#pragma unroll
  for (int e = 0; e < P; e++) wb[lb.bi[PL::NP - 1] + e] = v[e];
}

template <int N, int P, int PIDX, bool A>
RDSP_HD void inv_mid_load(const LdsBases<N, P, A> &lb, float2 *v, const float2 *wb) {
  constexpr int s = FftPlan<N, P>::span(PIDX);
#pragma unroll
  for (int j = 0; j < P; j++) v[j] = lds_ld(&wb[lb.bo[PIDX] + WbMap<N, P, A>::template off<PIDX>(j, s)]);
}
template <int N, int P, int PIDX, bool A>
RDSP_HD void inv_mid_store(const LdsBases<N, P, A> &lb, float2 *v, float2 *wb, const float2 *twp) {
  constexpr int s = FftPlan<N, P>::span(PIDX);
#pragma unroll
  for (int k = 1; k < P; k++) v[k] = cmulc(v[k], twp[k - 1]);
  Dft<P, true>::run(v);
#pragma unroll
  for (int j = 0; j < P; j++) wb[lb.bi[PIDX] + WbMap<N, P, A>::template off<PIDX - 1>(j, s)] = v[j];
}
template <int N, int P, int PIDX, bool A>
RDSP_HD void inv_pass_mid(const LdsBases<N, P, A> &lb, float2 *wb, const float2 *twp) {
  float2 v[P];
  inv_mid_load<N, P, PIDX, A>(lb, v, wb);
  inv_mid_store<N, P, PIDX, A>(lb, v, wb, twp);
}

/* pass 0 inverse: result v[j] = y[t + j*NT] */
template <int N, int P, bool A>
RDSP_HD void inv_pass0_load(const LdsBases<N, P, A> &lb, float2 *v, const float2 *wb, const float2 *twp) {
  using PL = FftPlan<N, P>;
#pragma unroll
  for (int j = 0; j < P; j++) v[j] = lds_ld(&wb[lb.bo[0] + WbMap<N, P, A>::template off<0>(j, PL::span(0))]);
#pragma unroll
  for (int k = 1; k < P; k++) v[k] = cmulc(v[k], twp[k - 1]);
  Dft<P, true>::run(v);
}

/* Twiddle storage of a thread.  CHAIN = false: all P-1 twiddles of every pass
 * in registers.  CHAIN = true: one base per pass, the rest regenerated by
 * twiddle_chain() right before the pass uses them. */
template <int N, int P, bool CHAIN>
struct Twiddles;

template <int N, int P>
struct Twiddles<N, P, false> {
  float2 tw[FftPlan<N, P>::NTW][P - 1];
  RDSP_HD void init(int t) { make_twiddles<N, P>(t, tw); }
  template <int PIDX>
  RDSP_HD void get(float2 *out) const {
#pragma unroll
    for (int k = 0; k < P - 1; k++) out[k] = tw[PIDX][k];
  }
};
template <int N, int P>
struct Twiddles<N, P, true> {
  float2 w1[FftPlan<N, P>::NTW];
  RDSP_HD void init(int t) { make_twiddle_bases<N, P>(t, w1); }
  template <int PIDX>
  RDSP_HD void get(float2 *out) const {
    float2 b = w1[PIDX];
#ifdef __HIP_DEVICE_COMPILE__
    /* opaque copy: stops loop-invariant code motion from hoisting the whole chain
     * out of the chunk loop (which would put all P-1 twiddles back in registers) */
    asm volatile("" : "+v"(b.x), "+v"(b.y));
#endif
    twiddle_chain<P>(b, out);
  }
};

/* compile-time loops over the middle passes */
template <int N, int P, int PIDX, int PEND, bool A, typename TW, typename SYNC>
RDSP_HD void fwd_mid_all(const LdsBases<N, P, A> &lb, float2 *wb, const TW &tw, SYNC sync) {
  if constexpr (PIDX < PEND) {
    float2 twp[P - 1];
    tw.template get<PIDX>(twp);
    fwd_pass_mid<N, P, PIDX, A>(lb, wb, twp);
    sync();
    fwd_mid_all<N, P, PIDX + 1, PEND, A>(lb, wb, tw, sync);
  }
}
template <int N, int P, int PIDX, bool A, typename TW, typename SYNC>
RDSP_HD void inv_mid_all(const LdsBases<N, P, A> &lb, float2 *wb, const TW &tw, SYNC sync) {
  if constexpr (PIDX >= 1) {
    float2 twp[P - 1];
    tw.template get<PIDX>(twp);
    inv_pass_mid<N, P, PIDX, A>(lb, wb, twp);
    sync();
    inv_mid_all<N, P, PIDX - 1, A>(lb, wb, tw, sync);
  }
}

}  // namespace rdsp

#endif
