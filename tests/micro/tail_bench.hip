// Times the tail kernel alone (ALS notch instance, 4096 channels by default).
// hipcc -O3 --offload-arch=gfx950 -std=c++17 -I radiodsp_sdr_rx_amd/csrc tests/micro/tail_bench.hip \
//       -L radiodsp_sdr_rx_amd -lrdsp_hip -Wl,-rpath,$PWD/radiodsp_sdr_rx_amd -o tests/micro/tail_bench
// usage: tail_bench [channels] [variant] [iterations]   variant 100 the product kernel (default; others: EXPERIMENTAL=1 builds)
#include "rdsp_kernels.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
int main(int argc, char **argv) {
  int nch = argc > 1 ? atoi(argv[1]) : 4096, nb = 128;
  const int variant = argc > 2 ? atoi(argv[2]) : 100;  // 100 row layout (default), 16 rdsp_tail.hip, 116 / 108 matrix pipe
  size_t stride = (size_t)nb * 128;
  float *mid, *w, *prev, *en, *scal; uint32_t *out;
  hipMalloc(&mid, nch * stride * 4); hipMalloc(&w, nch * 96 * 4); hipMalloc(&prev, nch * 128 * 4); hipMalloc(&en, nch * 4);
  hipMalloc(&scal, nch * 16); hipMalloc(&out, nch * stride * 4);
  std::vector<float> h(nch * stride);
  for (size_t i = 0; i < h.size(); i++) h[i] = 0.3f * sinf(0.05f * (float)(i % 9973)) + 0.01f * (float)((i * 2654435761u) % 1000) / 1000.f;
  hipMemcpy(mid, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(w, 0, nch * 96 * 4); hipMemset(prev, 0, nch * 128 * 4); hipMemset(en, 0, nch * 4); hipMemset(scal, 0, nch * 16);
  RdspTailParams p; memset(&p, 0, sizeof(p));
  p.mid = mid; p.mid_stride = stride; p.n_channels = nch; p.n_blocks = nb; p.als_mode = 1; p.als_mu = 0.02f; p.als_first = 1;
  p.als_w = w; p.als_prev = prev; p.als_energy = en; p.agc_on = 1; p.agc_attack = 0.6f; p.agc_decay = 0.03f; p.out_gain = 0.5f;
  p.st_scal = scal; p.out_i16 = out; p.out_stride = stride;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rdsp_launch_tail(&p, variant, 0); hipDeviceSynchronize();
  float best = 1e9;
  const int iters = argc > 3 ? atoi(argv[3]) : 5; /* many iterations: a run long enough to sample clock and power beside it */
  for (int it = 0; it < iters; it++) {
    hipEventRecord(e0); rdsp_launch_tail(&p, variant, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("EXP=%d nch=%d: %.3f ms for %d steps -> %.1f ns/step = %.0f clk@2.25GHz\n", 0, nch, best, nb * 128, best * 1e6 / (nb * 128), best * 1e6 / (nb * 128) * 2.25);
  return 0;
}
