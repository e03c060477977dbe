/*
 * rdsp_kernels.h -- parameter blocks shared by the HIP kernels and the C-ABI
 * launch layer (internal; the public boundary is include/rdsp.h).
 */
#ifndef RDSP_KERNELS_H
#define RDSP_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RDSP_BLOCK 128
#define RDSP_LMS_TAPS 96
#define RDSP_FD_N 512 /* transform size of the frequency-domain decimator (rdsp_front_fd_kernel) */
#define RDSP_FD_P 8
#define RDSP_XP 84 /* entries per polyphase sub-plane in LDS (81 used, 84 keeps
                      the sub-plane stride at 8 banks mod 32) */

enum { RDSP_K_DEMOD_IQ = 0, RDSP_K_DEMOD_REAL = 1, RDSP_K_DEMOD_AM = 2, RDSP_K_DEMOD_SAM = 3 };

/* One receiver group: channels of a group share tuning offset, filter
 * mask and demodulator.  128 bytes, read once per workgroup into scalar registers;
 * the host rewrites single records in stream order (rdsp_launch_group_store), so a
 * retune never stalls the processing stream. */
struct RdspGroup {
  uint32_t dphi;       /* NCO phase increment per sample (turns * 2^32)      */
  int32_t demod;       /* RDSP_K_DEMOD_*                                     */
  float2 rot1, rot2, rot3;    /* exp(-j*2pi*k*dphi/2^32), k = 1..3           */
  float2 rotp1, rotp2, rotp3; /* the same for k*4*NT samples (next load pass) */
  uint32_t mask_off;   /* float2 offset of the group's active mask in the pool */
  uint32_t dphi_hist;  /* increment the 256 history samples were mixed with: differs
                          from dphi only in the first call after a tuning change   */
  float2 roth1, roth2, roth3; /* rot1..3 for dphi_hist                            */
  float2 rothp3;       /* rotp3 for dphi_hist                                      */
  float2 rotq1, rotq2, rotq3; /* the same for 256 k samples: one quad column of the
                                 frequency-domain decimator's frame (rdsp_front_fd_kernel)   */
  uint32_t pad[2];
};

/* front kernel: A1 unpack, A2 mixer, A3 decimator, A5 overlap-save filter,
 * A6 spectral NR, demod select, and (when no NLMS stage follows) A9 AGC,
 * output gain and A10 pack. */
struct RdspFrontParams {
  const uint32_t *iq;  /* [ch][in_stride] int16 I | int16 Q << 16            */
  size_t in_stride;    /* samples per channel row                            */
  int n_chunks;        /* chunks of 256*DECIM input samples in this launch   */
  uint32_t n0;         /* absolute index (mod 2^32) of the first sample      */
  const RdspGroup *groups;   /* [n_groups] tuning / mask / demod records     */
  const uint16_t *group_of;  /* [ch] group of each channel; NULL: all group 0 */
  float scale_i, scale_q;  /* iq_balance*input_gain/32768, input_gain/32768  */
  int swap_iq;             /* preProcessor.swapIQ(true), INO:118             */
  /* what the previous call ran with: the FIR history is kept as raw words, and the 256 samples in
   * it went through the pre-processor and the input gains when they were new (like dphi_hist) */
  float scale_i_hist, scale_q_hist;
  int swap_hist;
  int nb_on;               /* noise blanker (engine feature, build-defined)  */
  float nb_thr;            /* blanking threshold as a power ratio            */
  const float *fir_hc;     /* [4][64] decimator taps, hc[c][k'] = h[4k'+c]   */
  const float2 *mask_pool; /* [n_groups][2][N] masks/N, digit-reversed, thread-major (double-buffered) */
  int spectral_on;         /* 1: SPEC:112-269 (smoothed floor); 2: the older variant of
                              backup/RadioDSP_SDR_RX_Conv.ino:1586-1630 (floor = threshold) */
  float spectral_k;        /* (float)(level*1.5); 3 for the older variant    */
  int spectral_literal;    /* 1: re-synthesis as SPEC:229-232 writes it: mag' (arm_cos_f32(phi) + j arm_sin_f32(phi)),
                              phi = atan2(im, re); 0: the exact-arithmetic equivalent X mag'/mag */
  const float *sin_table;  /* [513] sinTable_f32 of arm_sin_f32 / arm_cos_f32 (rdsp_arm_sin_table)  */
  int vad_lo, vad_hi;      /* inclusive natural bin range                    */
  int to_mid;              /* 1: write mono float audio for the tail kernel  */
  int lean;                /* 1: register-lean variant (FFT twiddles rebuilt per pass)     */
  int fir_matrix;          /* 1: decimating FIR as v_mfma GEMM slices (EXPERIMENTAL builds)    */
  int fir_fd;              /* decimator in the frequency domain (rdsp_front_fd_kernel): 1 = 448-sample frames anchored at
                              the call's first sample, 2 = frames of one granule (split-invariant); 0 = direct form;
                              on 16-lane rows (rdsp_front_rd_kernel): 3 = 128 outputs per 256-point window (two frames per
                              granule: split-invariant), 4 = 192 outputs per window, anchored at the call's first sample */
  const float2 *fd_mask;   /* [4][RDSP_FD_N] spectra of the polyphase branches g_r[k] = h[4k - r],
                              /RDSP_FD_N, digit-reversed thread-major like the filter masks    */
  const float2 *rd_mask;   /* [4][256] the same spectra for 256-point windows, natural bin order, /256 (fir_fd 3, 4) */
  int front_prio;          /* 1: raise wave priority (tail kernel shares the SIMDs) */
  int agc_on;
  float agc_attack, agc_decay;
  float out_gain;
  /* per-channel state */
  uint32_t *st_hist;       /* [ch][256] last raw input samples               */
  float2 *st_prev;         /* [ch][N/2] previous hop of the decimated stream */
  float *st_scal;          /* [ch][4]: NFloor, agc_g, am_dc, blanker level   */
  /* outputs */
  uint32_t *out_i16;       /* [ch][out_stride] int16 L | int16 R << 16       */
  size_t out_stride;
  float2 *out_f32;         /* optional [ch][out_stride] float L,R            */
  float *mid;              /* [ch][mid_stride] mono float (to_mid)           */
  float *mid_q;            /* [ch][mid_stride] Im y for SAM channels, else unused */
  size_t mid_stride;
  int ch_base;             /* first channel of this launch (workgroup b works on ch_base + b) */
};

/* SAM demodulator (PLL, serial in time): one channel per lane, in place on `mid`
 * for the channels whose group demodulates SAM */
struct RdspSamParams {
  float *mid;              /* in: Re y, out: audio                           */
  const float *mid_q;      /* Im y                                           */
  size_t mid_stride;
  int n_channels;
  int n_samples;           /* multiple of 32 (whole tiles)                   */
  const RdspGroup *groups;
  const uint16_t *group_of;
  float g1, g2, wmin, wmax;
  float *st_sam;           /* [ch][4]: phase, omega, loop filter output, dc  */
};

/* tail kernel: A7 NLMS noise reduction, A8 ALS notch/peak, A9 AGC, gain,
 * A10 pack.  One channel per LPC lanes. */
struct RdspTailParams {
  const float *mid;        /* [ch][mid_stride]                               */
  size_t mid_stride;
  int n_channels;          /* one past the last channel of this launch       */
  int ch_base;             /* first channel of this launch                   */
  int n_blocks;            /* 128-sample blocks at the decimated rate        */
  int nr_on, als_mode;     /* als_mode: 0 off, 1 notch (e), 2 peak (y)       */
  int nr_mode;             /* 0: 1.1*y (CONV:334), 2: plain y (NR:73)        */
  int prio;                /* wave priority of the tail kernel (s_setprio), 0..3 */
  int energy_running;      /* 1: arm_lms_norm_f32's energy as NR:73 runs it -- one running difference for the whole
                              stream; 0 (default): re-started from the exact window sum at every 128-sample block */
  float *raw_out;          /* non-null: write the stage output as floats
                              [ch][mid_stride] and skip AGC/gain/pack         */
  float nr_mu, als_mu;
  int nr_first, als_first; /* 1: first call ever (d = x quirk, NR:69-79)     */
  float *nr_w, *nr_prev, *nr_energy;    /* [ch][96], [ch][128], [ch]         */
  float *als_w, *als_prev, *als_energy;
  int agc_on;
  float agc_attack, agc_decay;
  float out_gain;
  float *st_scal;
  uint32_t *out_i16;
  size_t out_stride;
  float2 *out_f32;
  /* per-channel health words, sticky (OR-ed in at the end of a launch): [0 .. n) for the DSP-NR
   * instance, [n .. 2n) for the ALS instance, n = n_channels of the chain (st_status_stride);
   * bit 0: energy + eps <= 0 was seen (the step size of arm_lms_norm_f32 goes negative or infinite),
   * bit 1: a weight or the energy is not finite.  May be null. */
  uint32_t *st_status;
  size_t st_status_stride;
};

/* biquad cascades (rdsp_biquad.hip): four DF1 stages per channel, one stage per lane of a quad.
 * Float mode: `buf` [ch][stride] in place (the chain's mono intermediate).  int16 mode (in16 /
 * out16 non-null): samples taken / written every step16 / ostep16 int16 of [ch][stride16] rows. */
struct RdspBiquadParams {
  float *buf;
  size_t stride;
  const int16_t *in16;
  size_t stride16;
  int step16;
  int16_t *out16;
  size_t ostride16;
  int ostep16;
  int n_channels;          /* one past the last channel of this launch       */
  int ch_base;
  int n_samples;           /* multiple of 128                                */
  const float *coef;       /* [n_sets][20]: {b0,b1,b2,a1,a2} x 4, feedback terms added */
  const uint16_t *set_of;  /* [ch] coefficient set of each channel; NULL: set 0 */
  float *state;            /* [ch][4 stages][x1,x2,y1,y2]                    */
  /* AudioFilterBiquad objects (the Teensy library's fixed-point cascade; int16 in and out): non-NULL icoef selects it */
  const int *icoef;        /* [4][5] b0, b1, b2, -a1, -a2 scaled by 2^30      */
  int n_stages;            /* the cascade runs stages 0 .. n_stages - 1       */
  int *istate;             /* [ch][4 stages][x1, x2, y1, y2, sum]             */
};

#ifdef __cplusplus
extern "C" {
#endif
int rdsp_launch_biquad(const RdspBiquadParams *p, hipStream_t stream);
int rdsp_launch_biquad_coef_store(float *dst, const float *coef20, hipStream_t stream);
/* returns hipError_t as int */
int rdsp_launch_front(int fft_l, int decim, const RdspFrontParams *p, int n_channels,
                      hipStream_t stream);
int rdsp_launch_tail(const RdspTailParams *p, int lanes_per_channel, hipStream_t stream);
int rdsp_launch_sam(const RdspSamParams *p, hipStream_t stream);
int rdsp_launch_group_store(RdspGroup *dst, const RdspGroup *val, hipStream_t stream);
int rdsp_launch_iq_slip(const uint32_t *in, size_t in_stride, uint32_t *out, size_t out_stride, const uint32_t *carry_in,
                        size_t carry_stride, uint32_t *carry_out, int n_samples, int slip, int n_channels, hipStream_t stream);
int rdsp_launch_q15_to_float(const int16_t *src, float *dst, size_t n, hipStream_t stream);
int rdsp_launch_float_to_q15(const float *src, int16_t *dst, size_t n, hipStream_t stream);
size_t rdsp_front_lds_bytes(int fft_l, int decim);
#ifdef __cplusplus
}
#endif

#endif
