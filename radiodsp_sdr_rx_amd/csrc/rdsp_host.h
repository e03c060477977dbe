/*
 * rdsp_host.h -- internal host-side declarations shared by rdsp_design.c,
 * rdsp_chain.hip and rdsp_graph.c (the public boundary is include/rdsp.h).
 */
#ifndef RDSP_HOST_H
#define RDSP_HOST_H

#include "../../include/rdsp.h"

#ifdef __cplusplus
extern "C" {
#endif

int rdsp_plan_radix(int fft_l);
int rdsp_bin_of_pos(int fft_l, int i);
void rdsp_mask_device_image(const float *mask_nat, int fft_l, float *image);
int rdsp_design_decimator(int ntaps, double cut_hz, double fs, int window, float *h_nat, float *hc);
int rdsp_fd_decimator_image(const float *h_nat, int fft_l, float *image);
int rdsp_rd_decimator_image(const float *h_nat, float *image);
uint32_t rdsp_nco_dphi(double hz, double fs);
void rdsp_nco_rot(uint32_t dphi, int k, float *out2);
float rdsp_lms_mu(int strength);
void rdsp_sam_constants(double fs_out, float *g1, float *g2, float *wmin, float *wmax);
void rdsp_set_error(const char *fmt, ...);
void rdsp_arm_sin_table(float *tab513);          /* sinTable_f32 of arm_sin_f32 / arm_cos_f32 */
void rdsp_q15_twiddles(int n, uint32_t *out);   /* [3n/4] cos | sin << 16 of 2 pi m / n, twiddleCoef_4096_q15's rule */
const uint16_t *rdsp_sqrt_guess_table(void);   /* [33] */
void rdsp_host_fft(double *re, double *im, int n); /* in-place radix-2 forward transform, n a power of two */

#ifdef __cplusplus
}
#endif
#endif
