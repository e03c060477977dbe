// Host emulation of the polyphase LDS layout + fir_lane (rdsp_front.h) against a
// float64 direct-form decimating FIR:  y[m] = sum_k h[k] x[4m-k].
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "rdsp_front.h"
using namespace rdsp;
int main() {
  std::vector<float2> xs(16 * RDSP_XP, make_float2(1e30f, 1e30f));  // poison
  std::vector<float2> x(256 + 2048);  // x[n], n in [-256, 2047], two chunks
  std::vector<float> h(256), hc(256);
  srand(7);
  for (auto &v : x) v = make_float2(rand() / (float)RAND_MAX - .5f, rand() / (float)RAND_MAX - .5f);
  for (auto &v : h) v = rand() / (float)RAND_MAX - .5f;
  for (int c = 0; c < 4; c++) for (int k = 0; k < 64; k++) hc[c * 64 + k] = h[4 * k + c];
  double worst = 0;
  for (int n = -256; n < 0; n++) xs[xs_pos(n)] = x[n + 256];
  for (int chunk = 0; chunk < 2; chunk++) {
    for (int n = 0; n < 1024; n++) xs[xs_pos(n)] = x[256 + chunk * 1024 + n];
    for (int l = 0; l < 64; l++) {
      float2 acc[4] = {};
      fir_lane(l, 0, 2, xs.data(), reinterpret_cast<const float4 *>(hc.data()), acc);   // split like the 4-wave variant
      fir_lane(l, 2, 4, xs.data(), reinterpret_cast<const float4 *>(hc.data()), acc);
      for (int r = 0; r < 4; r++) {
        int m = 4 * l + r;
        double sr = 0, si = 0;
        for (int k = 0; k < 256; k++) {
          int n = chunk * 1024 + 4 * m - k;
          sr += (double)h[k] * x[256 + n].x;
          si += (double)h[k] * x[256 + n].y;
        }
        worst = fmax(worst, fmax(fabs(sr - acc[r].x), fabs(si - acc[r].y)));
      }
    }
    { float4 *x4 = reinterpret_cast<float4 *>(xs.data()); for (int i = 0; i < 8 * 17; i++) { int sp = i / 17, e = i % 17; x4[sp * RDSP_XP + e] = x4[sp * RDSP_XP + 64 + e]; } }
  }
  printf("fir max abs err %.3e\n", worst);
  // NCO phasor accuracy (ALU version used by the kernel)
  double pw = 0;
  double pa = 0;
  for (int i = 0; i < 400000; i++) {
    uint32_t ph = (uint32_t)rand() * 2654435761u + (uint32_t)i * 7919u;
    float2 p = nco_phasor_alu(ph);
    double th = 2 * M_PI * ph / 4294967296.0;
    pa = fmax(pa, fmax(fabs(p.x - cos(th)), fabs(p.y + sin(th))));
  }
  printf("nco alu max abs err %.3e\n", pa);
  if (pa > 3e-7) { printf("FAIL\n"); return 1; }
  if (worst > 2e-5 || pw > 3e-7) { printf("FAIL\n"); return 1; }
  printf("OK\n");
  return 0;
}
