/*
 * rdsp_sam.hip -- SAMmode (CTL:384-391): the AudioSDR engine's synchronous AM detector, a
 * second-order PLL on the filtered base band (arithmetic build-defined,
 * DESIGN.md 6e).  Runs between the front kernel and the tail stage for the channels of
 * SAM groups.
 */
#include "rdsp_wave.h"

using namespace rdsp;

namespace {

/* ---- SAM demodulator: second-order PLL on the filtered base band --------------------
 * SAMmode (CTL:384-391) is an AudioSDR demodulator, build-defined here:
 *   corr0 = Re(y e^{-j phs}), corr1 = Im(y e^{-j phs}), det = atan2(corr1, corr0) weighted
 *   by |y|^2/(|y|^2 + 1e-6), omega += g2 det (clamped), phs += previous (g1 det + omega),
 *   audio = corr0 - dc.
 * The recursion is serial per sample, so a channel is one lane (64 channels per wave, 64 waves
 * at 4096 channels) and the kernel is bound by the length of the dependent chain, not by
 * issue slots or bandwidth.  What keeps the chain short:
 *  - the oscillator is a unit phasor rotated by the loop filter output instead of sincos(phs).
 *    The phase update uses the *previous* sample's filter output, so the phasor of sample n+1
 *    depends on the detector of sample n-1, not n: the recursion is two interleaved chains
 *    (even and odd samples) joined only by the three-operation omega update, and the eight
 *    samples of a loop body give the scheduler both chains to overlap;
 *  - the detector is an arctangent without a range fold: one reciprocal, an eight-term odd polynomial
 *    on [0, 1], three selects for the octant, 1.5e-7;
 *  - the unit-modulus correction of the phasor (once per eight samples) takes its norm from the
 *    previous sample's phasor, off the chain;
 *  - samples reach the lanes through LDS tiles: the global accesses are 16-byte and
 *    coalesced (eight lanes per 128-byte row piece), the next tile's loads are in flight
 *    while this one is computed.  One lane reading its own row straight from HBM touches
 *    64 different cache lines per load instruction.
 * phs is rebuilt from the phasor when the state is saved. */
__device__ __forceinline__ float sam_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(fmaxf(ax, ay), 1e-30f), mn = fminf(ax, ay);
  /* atan(u), u = mn / mx in [0, 1], as u P(u^2) with an eight-term polynomial (weighted least squares
   * towards the minimax fit, 1.5e-7 in float): no range fold, so no selects and one reciprocal */
  const float u = mn * __builtin_amdgcn_rcpf(mx);
  const float z = u * u;
  constexpr float c[8] = {9.999993356e-01f, -3.332986079e-01f, 1.994656566e-01f, -1.390862957e-01f, 9.642197347e-02f, -5.591232664e-02f, 2.186295759e-02f, -4.054567096e-03f};
  float p = c[7];
#pragma unroll
  for (int k = 6; k >= 0; k--) p = fmaf(p, z, c[k]);
  float r = p * u;
  r = (ay > ax) ? 1.57079632679489662f - r : r;
  r = (x < 0.0f) ? 3.14159265358979324f - r : r;
  return (y < 0.0f) ? -r : r;
}
/* sine and cosine of a small angle (|d| < 0.6 rad: the loop filter output is clamped
 * to +-2 kHz at 24 kHz plus g1*pi), Taylor to d^9 / d^8: 1e-9 */
__device__ __forceinline__ void sam_sincos_small(float d, float *sn, float *cs) {
  const float d2 = d * d;
  float s = fmaf(d2, 2.7557319e-6f, -1.9841270e-4f);
  s = fmaf(s, d2, 8.3333333e-3f);
  s = fmaf(s, d2, -1.6666667e-1f);
  *sn = fmaf(s * d2, d, d);
  float c = fmaf(d2, 2.4801587e-5f, -1.3888889e-3f);
  c = fmaf(c, d2, 4.1666667e-2f);
  c = fmaf(c, d2, -0.5f);
  *cs = fmaf(c, d2, 1.0f);
}

typedef float sam_v4 __attribute__((ext_vector_type(4)));
constexpr int SAM_TILE = 32;           /* samples per tile and channel                         */
constexpr int SAM_ROW = SAM_TILE + 4;  /* row pitch in floats: 16-byte aligned, 4 banks skew   */

__global__ void __launch_bounds__(64) rdsp_sam_kernel(RdspSamParams p) {
  __shared__ __attribute__((aligned(16))) float ti[64 * SAM_ROW];
  __shared__ __attribute__((aligned(16))) float tq[64 * SAM_ROW];
  const int lane = threadIdx.x;
  const size_t ch0 = (size_t)blockIdx.x * 64;
  const size_t ch = ch0 + lane;
  const bool in_range = ch < (size_t)p.n_channels;
  bool is_sam = false;
  if (in_range) {
    const uint32_t gi = p.group_of ? (uint32_t)p.group_of[ch] : 0u;
    is_sam = p.groups[gi].demod == RDSP_K_DEMOD_SAM;
  }
  const unsigned long long sam_mask = __ballot(is_sam);
  if (sam_mask == 0ull) return; /* nothing to do for these 64 channels (uniform) */

  /* tile transfers: lane takes the 16-byte piece `piece` of rows row0 + 8 j, j < 8 */
  const int piece = lane & 7, row0 = lane >> 3;
  const size_t last = (size_t)p.n_channels - 1;
  const float *src_i[8], *src_q[8]; /* rows past the last channel read a real one, nothing is stored for them */
#pragma unroll
  for (int j = 0; j < 8; j++) {
    size_t c = ch0 + row0 + 8 * j;
    c = c < last ? c : last;
    src_i[j] = p.mid + c * p.mid_stride + 4 * piece;
    src_q[j] = p.mid_q + c * p.mid_stride + 4 * piece;
  }
  float omega = 0.f, fil = 0.f, dc = 0.f, cs = 1.f, sn = 0.f;
  if (is_sam) {
    omega = p.st_sam[ch * 4 + 1];
    fil = p.st_sam[ch * 4 + 2];
    dc = p.st_sam[ch * 4 + 3];
    sincosf(p.st_sam[ch * 4 + 0], &sn, &cs);
  }
  float nrm = 1.0f; /* unit-modulus correction from the previous phasor */
  sam_v4 ri[8], rq[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    ri[j] = *reinterpret_cast<const sam_v4 *>(src_i[j]);
    rq[j] = *reinterpret_cast<const sam_v4 *>(src_q[j]);
  }
  float *mine_i = ti + lane * SAM_ROW;
  const float *mine_q = tq + lane * SAM_ROW;
#pragma unroll 1
  for (int n0 = 0; n0 < p.n_samples; n0 += SAM_TILE) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      *reinterpret_cast<sam_v4 *>(ti + (row0 + 8 * j) * SAM_ROW + 4 * piece) = ri[j];
      *reinterpret_cast<sam_v4 *>(tq + (row0 + 8 * j) * SAM_ROW + 4 * piece) = rq[j];
    }
    wg_sync<1>();
    if (n0 + SAM_TILE < p.n_samples) { /* the next tile lands during this tile's arithmetic */
#pragma unroll
      for (int j = 0; j < 8; j++) {
        ri[j] = *reinterpret_cast<const sam_v4 *>(src_i[j] + n0 + SAM_TILE);
        rq[j] = *reinterpret_cast<const sam_v4 *>(src_q[j] + n0 + SAM_TILE);
      }
    }
#pragma unroll 1
    for (int m = 0; m < SAM_TILE; m += 8) {
      float I[8], Q[8], o[8];
      {
        const float4 a = *reinterpret_cast<const float4 *>(mine_i + m), b = *reinterpret_cast<const float4 *>(mine_i + m + 4);
        const float4 c = *reinterpret_cast<const float4 *>(mine_q + m), d = *reinterpret_cast<const float4 *>(mine_q + m + 4);
        I[0] = a.x; I[1] = a.y; I[2] = a.z; I[3] = a.w; I[4] = b.x; I[5] = b.y; I[6] = b.z; I[7] = b.w;
        Q[0] = c.x; Q[1] = c.y; Q[2] = c.z; Q[3] = c.w; Q[4] = d.x; Q[5] = d.y; Q[6] = d.z; Q[7] = d.w;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        /* the oscillator for the next sample first: phs += fil with fil from the previous
         * sample, so this rotation does not wait for this sample's detector */
        float sd, cd;
        sam_sincos_small(fil, &sd, &cd);
        if (k == 7) { /* one Newton step towards |.| = 1 per eight rotations (drift per rotation ~6e-8),
                         with the norm of the phasor of a sample earlier: off the chain */
          sd *= nrm;
          cd *= nrm;
        }
        const float c1 = cs * cd - sn * sd, s1 = sn * cd + cs * sd;
        if (k == 6) nrm = fmaf(-0.5f, c1 * c1 + s1 * s1, 1.5f);
        const float corr0 = I[k] * cs + Q[k] * sn;
        const float corr1 = Q[k] * cs - I[k] * sn;
        const float mag2 = corr0 * corr0 + corr1 * corr1;
        const float det = sam_atan2(corr1, corr0) * (mag2 * __builtin_amdgcn_rcpf(mag2 + 1e-6f));
        omega = __builtin_amdgcn_fmed3f(omega + p.g2 * det, p.wmin, p.wmax); /* clamp */
        fil = p.g1 * det + omega;
        cs = c1;
        sn = s1;
        dc = dc + (corr0 - dc) * (1.0f / 512.0f);
        o[k] = corr0 - dc;
      }
      *reinterpret_cast<float4 *>(mine_i + m) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4 *>(mine_i + m + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    wg_sync<1>();
    { /* audio of the SAM channels back in place; rows of other demodulators keep their base band */
      const int n = n0 + 4 * piece;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int r = row0 + 8 * j;
        if ((sam_mask >> r) & 1ull)
          *reinterpret_cast<float4 *>(p.mid + (ch0 + r) * p.mid_stride + n) =
              *reinterpret_cast<const float4 *>(ti + r * SAM_ROW + 4 * piece);
      }
    }
    wg_sync<1>();
  }
  if (is_sam) {
    float phs = atan2f(sn, cs);
    if (phs < 0.0f) phs += 6.28318530717958647692f;
    p.st_sam[ch * 4 + 0] = phs;
    p.st_sam[ch * 4 + 1] = omega;
    p.st_sam[ch * 4 + 2] = fil;
    p.st_sam[ch * 4 + 3] = dc;
  }
}

}  // namespace

extern "C" int rdsp_launch_sam(const RdspSamParams *p, hipStream_t stream) {
  /* whole tiles: the chain hands over n_blocks * 128 / decim samples, a multiple of 32 */
  if (p->n_samples <= 0 || p->n_samples % SAM_TILE != 0 || p->n_channels <= 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(rdsp_sam_kernel, dim3((p->n_channels + 63) / 64), dim3(64), 0, stream, *p);
  return (int)hipGetLastError();
}
