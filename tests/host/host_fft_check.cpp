// Host-side emulation of the workgroup FFT (rdsp_fft.h): runs every "thread" in
// sequence, with the kernel's __syncthreads() points as loop boundaries, and
// checks the forward result (at digit-reversed positions) and the round trip
// against a float64 DFT.  Build: hipcc -O2 -I radiodsp_sdr_rx_amd/csrc ...
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <complex>
#include <type_traits>
#include "rdsp_fft.h"
using namespace rdsp;

template <int N, int P, bool CHAIN, bool ALIAS = false>
double check() {
  using PL = FftPlan<N, P>;
  constexpr int NT = PL::NT;
  // ALIAS: the work buffer lives inside the 8 polyphase planes (16*XP float2), behind
  // the 17-entry history of each plane, which is poisoned here and must stay intact
  std::vector<float2> x(N), wb(ALIAS ? 16 * RDSP_XP : PL::WB, make_float2(7e33f, 7e33f));
  srand(N * 31 + P);
  for (auto &v : x) v = make_float2(rand() / (float)RAND_MAX - 0.5f, rand() / (float)RAND_MAX - 0.5f);
  std::vector<std::complex<double>> X(N);
  for (int k = 0; k < N; k++) {
    std::complex<double> a = 0;
    for (int n = 0; n < N; n++)
      a += std::complex<double>(x[n].x, x[n].y) * std::polar(1.0, -2.0 * M_PI * (double)((long)k * n % N) / N);
    X[k] = a;
  }
  static float2 tw[NT][PL::NTW][P - 1];
  for (int t = 0; t < NT; t++) {
    if (CHAIN) {
      float2 w1[PL::NTW];
      make_twiddle_bases<N, P>(t, w1);
      for (int p = 0; p < PL::NTW; p++) twiddle_chain<P>(w1[p], tw[t][p]);
    } else {
      make_twiddles<N, P>(t, tw[t]);
    }
  }
  static LdsBases<N, P, ALIAS> lb[NT];
  for (int t = 0; t < NT; t++) make_lds_bases<N, P, ALIAS>(t, lb[t]);
  auto nosync = []() {};
  // forward pass 0
  for (int t = 0; t < NT; t++) {
    float2 v[P];
    for (int j = 0; j < P; j++) v[j] = x[t + j * NT];
    fwd_pass0_store<N, P>(lb[t], v, wb.data(), tw[t][0]);
  }
  // middle passes: emulated pass by pass over all threads.  The one-wave plans read under the map of one
  // exchange and write under the next one's (every lane of the wave has read before any writes): all
  // loads of a pass first, then all stores.  The four-wave plans run in place, thread after thread.
  static float2 regs[NT][P];
  auto fwd_mid = [&](auto pidx) {
    constexpr int PIDX = decltype(pidx)::value;
    if constexpr (PL::PERX) {
      for (int t = 0; t < NT; t++) fwd_mid_load<N, P, PIDX, ALIAS>(lb[t], regs[t], wb.data());
      for (auto &e : wb) if (!ALIAS) e = make_float2(7e33f, 7e33f);  // nothing of the old image may be read again
      for (int t = 0; t < NT; t++) fwd_mid_store<N, P, PIDX, ALIAS>(lb[t], regs[t], wb.data(), tw[t][PIDX]);
    } else {
      for (int t = 0; t < NT; t++) fwd_pass_mid<N, P, PIDX, ALIAS>(lb[t], wb.data(), tw[t][PIDX]);
    }
  };
  auto inv_mid = [&](auto pidx) {
    constexpr int PIDX = decltype(pidx)::value;
    if constexpr (PL::PERX) {
      for (int t = 0; t < NT; t++) inv_mid_load<N, P, PIDX, ALIAS>(lb[t], regs[t], wb.data());
      for (auto &e : wb) if (!ALIAS) e = make_float2(7e33f, 7e33f);
      for (int t = 0; t < NT; t++) inv_mid_store<N, P, PIDX, ALIAS>(lb[t], regs[t], wb.data(), tw[t][PIDX]);
    } else {
      for (int t = 0; t < NT; t++) inv_pass_mid<N, P, PIDX, ALIAS>(lb[t], wb.data(), tw[t][PIDX]);
    }
  };
  if constexpr (PL::NP >= 3) fwd_mid(std::integral_constant<int, 1>());
  if constexpr (PL::NP >= 4) fwd_mid(std::integral_constant<int, 2>());
  if constexpr (PL::NP >= 5) fwd_mid(std::integral_constant<int, 3>());
  static_assert(PL::NP <= 5, "extend harness");
  std::vector<float2> spec(N);
  double emax = 0, xmax = 0;
  for (int t = 0; t < NT; t++) {
    float2 v[P];
    fwd_pass_last<N, P>(lb[t], v, wb.data());
    for (int e = 0; e < P; e++) {
      int pos = t * P + e, k = bin_of_pos<N, P>(pos);
      spec[pos] = v[e];
      double d = std::abs(std::complex<double>(v[e].x, v[e].y) - X[k]);
      if (d > emax) emax = d;
      if (std::abs(X[k]) > xmax) xmax = std::abs(X[k]);
    }
  }
  double fwd_err = emax / xmax;
  // inverse
  for (int t = 0; t < NT; t++) {
    float2 v[P];
    for (int e = 0; e < P; e++) v[e] = spec[t * P + e];
    inv_pass_last<N, P>(lb[t], v, wb.data());
  }
  if constexpr (PL::NP >= 5) inv_mid(std::integral_constant<int, 3>());
  if constexpr (PL::NP >= 4) inv_mid(std::integral_constant<int, 2>());
  if constexpr (PL::NP >= 3) inv_mid(std::integral_constant<int, 1>());
  double imax = 0;
  for (int t = 0; t < NT; t++) {
    float2 v[P];
    inv_pass0_load<N, P>(lb[t], v, wb.data(), tw[t][0]);
    for (int j = 0; j < P; j++) {
      int n = t + j * NT;
      double d = std::hypot(v[j].x / (double)N - x[n].x, v[j].y / (double)N - x[n].y);
      if (d > imax) imax = d;
    }
  }
  (void)nosync;
  if (ALIAS) {
    for (int pl = 0; pl < 8; pl++)
      for (int e = 0; e < 34; e++)
        if (wb[pl * 2 * RDSP_XP + e].x != 7e33f) { printf("history entry clobbered: plane %d float2 %d\n", pl, e); return 1.0; }
  }
  printf("N=%d P=%d NT=%d NP=%d RL=%d alias=%d chain=%d fwd_err=%.3e roundtrip_err=%.3e\n", N, P, NT, PL::NP, PL::RL, (int)ALIAS, (int)CHAIN, fwd_err, imax);
  return fwd_err > imax ? fwd_err : imax;
}

template <int N, int P>
void print_maps() {
  using PL = FftPlan<N, P>;
  for (int x = 0; x < PL::NP - 1; x++) printf("map %d %d %d %d %d %d\n", N, P, x, PL::xshift(x), PL::xmul(x), PL::WB);
}

int main(int argc, char **argv) {
  if (argc > 1) { /* the LDS maps of the plans, for tests/micro/lds_model.py (test_host_logic.py) */
    print_maps<256, 4>(); print_maps<512, 8>(); print_maps<1024, 16>(); print_maps<2048, 8>(); print_maps<4096, 16>();
    return 0;
  }
  double w = 0;
  w = fmax(w, check<256, 4, false>());
  w = fmax(w, check<512, 8, false>());
  w = fmax(w, check<1024, 16, false>());
  w = fmax(w, check<2048, 8, false>());
  w = fmax(w, check<4096, 16, false>());
  w = fmax(w, check<256, 4, true>());
  w = fmax(w, check<512, 8, true>());
  w = fmax(w, check<1024, 16, true>());
  w = fmax(w, check<2048, 8, true>());
  w = fmax(w, check<4096, 16, true>());
  w = fmax(w, check<256, 4, false, true>());
  w = fmax(w, check<512, 8, false, true>());
  w = fmax(w, check<512, 8, true, true>());
  if (w > 2e-6) { printf("FAIL %.3e\n", w); return 1; }
  printf("OK\n");
  return 0;
}
