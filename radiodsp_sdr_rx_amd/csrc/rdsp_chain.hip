/*
 * rdsp_chain.hip -- C-ABI launch layer: the chain object (per-channel state in
 * HBM, shared tables) and the calls declared in include/rdsp.h.  Host logic
 * only; the arithmetic lives in rdsp_kernels.hip.  No CPU fallback exists: if
 * HIP reports no device every compute entry point returns RDSP_ERR_NO_DEVICE.
 */
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rdsp_host.h"
#include "rdsp_kernels.h"

static thread_local char g_err[512] = "";

extern "C" void rdsp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *rdsp_last_error(void) { return g_err; }
#ifdef RDSP_EXPERIMENTAL
extern "C" const char *rdsp_version(void) { return "rdsp-amd 0.2 (gfx950, experimental variants)"; }
extern "C" int rdsp_experimental_build(void) { return 1; }
#else
extern "C" const char *rdsp_version(void) { return "rdsp-amd 0.2 (gfx950)"; }
extern "C" int rdsp_experimental_build(void) { return 0; }
#endif

extern "C" int rdsp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

#define HIP_TRY(expr)                                                              \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      rdsp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RDSP_ERR_HIP;                                                         \
    }                                                                              \
  } while (0)

/* host side of one receiver group: filter, tuning offset, demodulator.
 * The device sees it as one RdspGroup record plus two mask buffers; a retune fills
 * the buffer the record does not point to (copy stream) and the record is rewritten
 * in stream order at the next processing call. */

struct GroupState {
  double lo = 0.0, hi = 0.0, nco_hz = 0.0;
  int demod = RDSP_DEMOD_USB, audio_filter = RDSP_AUDIO_2700;
  std::vector<double> coef_I, coef_Q; /* FIR_Coef_I/Q, CONV:69-70 */
  std::vector<float> mask_nat;        /* FIR_filter_mask, CONV:77 */
  int applied = 0;   /* mask buffer the device record points to (as queued) */
  int staged = -1;   /* buffer holding a newer mask that is not yet switched in */
  bool dirty = true; /* the device record must be rewritten before the next launch */
  bool has_dev_dphi = false; /* dev_dphi: increment the last launch mixed with */
  uint32_t dev_dphi = 0;
  /* two pinned images of the mask (N float2 each) with the event of the upload that last read each:
   * a retune fills the one whose upload is two retunes old, so the host never overwrites an image
   * an earlier, still queued upload has yet to read (the processing stream only waits for uploads
   * on the device; the host does not know when one has run) */
  hipEvent_t ev_copy[2] = {nullptr, nullptr};
  float *staging[2] = {nullptr, nullptr};
  bool issued[2] = {false, false};
  int stage_next = 0, stage_last = 0;
  float iir[20];            /* the group's audio band-pass as four biquads (RDSP_AUDIO_KIND_IIR) */
  bool iir_dirty = true;
};

struct rdsp_chain {
  rdsp_chain_config_t cfg;
  int n_channels, device, max_blocks;
  int N, decim, hop;
  uint64_t n_in;       /* absolute input sample counter */
  int old_nr_level;    /* oldNRLevel, CONV:80 */
  float nr_mu, als_mu;
  long nr_calls, als_calls; /* NR:69 ring statics: only "first call" matters */
  std::vector<float> fir_nat;
  std::vector<GroupState> groups;     /* at least one */
  std::vector<uint16_t> group_of;     /* empty: every channel in group 0 */
  /* device */
  RdspGroup *d_groups = nullptr;
  uint16_t *d_group_of = nullptr;
  float2 *d_mask_pool = nullptr;      /* [n_groups][2][N] */
  hipStream_t s_copy = nullptr;       /* mask uploads, concurrent with processing */
  hipEvent_t ev_fence = nullptr;      /* after the most recent front launch */
  bool fence_valid = false;
  float *d_fir_hc = nullptr;
  float2 *d_fd_mask = nullptr; /* [4][512] branch spectra of the frequency-domain decimator (decim 4 only) */
  float2 *d_rd_mask = nullptr; /* [4][256] the same for the 256-point windows of the row forms */
  float *d_sin_table = nullptr; /* [513] sinTable_f32 (spectral stage as written, rdsp_set_spectral_resynthesis); made on first use */
  int spectral_literal = 0;
  /* rdsp_sdr_set_engine_literal: the reference's own pre-processor and engine in front of the CONV stage (INO:53-54,71-86) */
  rdsp_engine_t *engine = nullptr;
  rdsp_preproc_t *pre = nullptr;
  int16_t *d_engine_io = nullptr; /* [n_channels][max_blocks * 128][2]: what the record queues would hold */
  int nlms_energy_running = 0; /* rdsp_set_nlms_energy_mode */
  uint32_t *d_hist = nullptr;
  float2 *d_prev = nullptr;
  float *d_scal = nullptr;
  float *d_nr_w = nullptr, *d_nr_prev = nullptr, *d_nr_energy = nullptr;
  float *d_als_w = nullptr, *d_als_prev = nullptr, *d_als_energy = nullptr;
  uint32_t *d_status = nullptr; /* [2][n_channels] sticky NLMS health words: DSP-NR instance, ALS instance */
  float *d_mid = nullptr;
  size_t mid_stride = 0;
  /* SAM groups: quadrature part of the base band (three buffers, like d_mid), PLL state */
  float *d_mid_q[3] = {nullptr, nullptr, nullptr}, *d_sam = nullptr;
  /* pipelined mode: the serial tail stage of call k runs on an internal stream,
   * concurrently with the front stage of call k+1 (three intermediate buffers: the front
   * stage of call k+1 never waits for the tail stage of call k-1) */
  int pipe_on = 0;
  hipStream_t s_tail = nullptr;
  /* the serial per-channel stages between front and tail stage (SAM PLL, IIR cascade) run on a stream
   * of their own when pipelined: three stages in flight, the tail stage waits for ev_mid */
  hipStream_t s_mid = nullptr;
  hipEvent_t ev_mid[3] = {nullptr, nullptr, nullptr};
  /* three intermediate buffers: the front stage may run two calls ahead of the tail stage, so
   * neither stream waits on the other in steady state (with two, every call paid two
   * cross-stream event waits, ~0.1 ms of a 2 ms step) */
  hipEvent_t ev_front[3] = {nullptr, nullptr, nullptr}, ev_tail[3] = {nullptr, nullptr, nullptr}, ev_misc = nullptr;
  /* pipelined calls over many channels go out as channel sub-batches: front(A), front(B), ... on the
   * caller's stream, tail(A), tail(B), ... on the tail stream, tail(A) waiting for front(A) only.
   * Every launch then has the shape the kernels' co-residency was balanced for (one tail wave and
   * two front waves per SIMD at 4096 channels), and the halves of one call overlap each other. */
  int sub_batch = 4096;
  std::vector<hipEvent_t> ev_front_sb[3]; /* [slot][sub-batch], created on first use */
  float *d_midx[2] = {nullptr, nullptr}; /* slots 1 and 2 (slot 0 is d_mid) */
  long call_idx = 0;
  int tail_slot = -1; /* slot of the last call whose tail stage went to s_tail (its ev_tail marks
                         when d_out, the AGC gain and the NLMS state of that call are final); -1: none */
  /* optional per-kernel HIP-event timing (bench.py roofline leg) */
  int timing_on = 0;
  std::vector<hipEvent_t> ev; /* pool, groups of 4: front begin/end, tail begin/end */
  std::vector<int> ev_has_tail;
  size_t ev_used = 0; /* calls recorded so far */
  int lean_mode = -1; /* -1 auto (= full), 0 full-register front kernel, 1 lean */
  int fir_mode = -1;  /* stage A3 (rdsp_chain_set_fir_variant): 4 frequency domain, one granule per frame (split-
                         invariant bits); -1 (default) that or 5, by what follows the front kernel; 0 direct form; 2
                         frequency domain, 448-sample frames;
                         5 / 6 frequency domain on 16-lane rows, 128 (split-invariant) / 192 outputs per 256-point window;
                         EXPERIMENTAL builds: 1 matrix-core FIR, 3 matrix unless the tail stage shares the SIMDs */
  /* wave priorities while both kernels share the SIMDs: the direct-form front kernel raises its
   * own to front_fir_prio during the FIR, the frequency-domain one never does; the tail kernel runs
   * at tail_prio throughout.  Round 2, frequency-domain front kernel, tail priority 0 / 1 / 2 / 3:
   * K3 1.191 / - / 1.188 / - ms, K5 2.72 / 2.36 / 2.34 / 2.36 ms per step (at equal priority the tail
   * kernels of two sub-batches are starved by the front waves).  Round 5 looked at the library's default decimator
   * (one granule per frame: half as much front-kernel work again per step), where the tail kernel is the starved one
   * (1.4 - 2.0 ms per launch against 1.06 alone): tail priority 0 instead of 2 measured 1.335-1.513 against 1.423-1.689
   * ms per K3 step in one interleaved A/B, 1.387-1.472 against 1.465-1.543 in a second, and 1.95 against 1.63 under
   * the profiler and 1.83 against 1.50 as a leg of the default bench run -- no consistent gain, so the priority stays 2
   * in every form (tests/micro/prio_default.sh, default_form_trace.sh) */
  int front_fir_prio = 2, tail_prio = 2;
  /* tail kernel: 100 = the product's (rdsp_tail.hip: a channel per 16-lane DPP row, two steps per reduction);
   * other values select the EXPERIMENTAL=1 variants (rdsp_launch_tail) */
  int tail_lpc = 100;
  int saved_agc_mode = RDSP_AGC_MEDIUM, saved_als_mode = RDSP_ALS_NOTCH;
  /* the engine's IIR audio filter bank (RDSP_AUDIO_KIND_IIR): coefficient sets per group, DF1
   * state per channel; allocated by rdsp_sdr_setAudioFilterKind */
  const char *front_name = "rdsp_front_kernel"; /* front kernel of the most recent call (measurement reports) */
  int audio_kind = RDSP_AUDIO_KIND_MASK;
  float *d_iir_coef = nullptr, *d_iir_state = nullptr;
  int iir_sets = 0;
  int swap_iq = 0;            /* preProcessor.swapIQ, INO:118 */
  int iq_slip = 0;            /* rdsp_pre_setIQslip: +1 delays the I rail by one sample, -1 the Q rail */
  uint32_t *d_slip_buf = nullptr;      /* [n_channels][max_blocks * 128] corrected words of a call */
  uint32_t *d_slip_carry[2] = {nullptr, nullptr}; /* [n_channels] last raw word of the previous / this call */
  int slip_phase = 0;
  bool slip_prev_on = false;  /* the previous call ran with the correction (its history words are corrected ones) */
  /* swap flag and input scales of the previous call (its samples are this call's FIR history) */
  bool hist_valid = false;
  int hist_swap = 0;
  float hist_scale_i = 0.f, hist_scale_q = 0.f;
  int nb_on = 0;              /* SDR.enableNoiseBlanker, BK_INO:1259 */
  float nb_threshold_db = 10.0f;
};

static int drain_tail_fwd(rdsp_chain_t *c);
static int ensure_sam(rdsp_chain_t *c);
static int ensure_sub_batch_events(rdsp_chain_t *c);
static void passband(int filter, int demod, double *lo, double *hi);
static int chain_build(rdsp_chain_t *c, const rdsp_chain_config_t *cfg, int n_channels, int device,
                       int max_blocks_per_call, int decim);
static int check_device(rdsp_chain_t *c) {
  if (hipSetDevice(c->device) != hipSuccess) {
    rdsp_set_error("hipSetDevice(%d) failed", c->device);
    return RDSP_ERR_HIP;
  }
  return RDSP_OK;
}

static void agc_params(int mode, float *attack, float *decay) {
  *attack = 0.6f;
  switch (mode) {
    case RDSP_AGC_FAST: *decay = 0.10f; break;
    case RDSP_AGC_MEDIUM: *decay = 0.03f; break;
    case RDSP_AGC_SLOW: *decay = 0.008f; break;
    default: *decay = 0.0f; break;
  }
}

static uint32_t demod_tuning_offset(int demod) {
  /* `TuningOffset = SDR.setDemodMode(mode)` (INO:139, CTL:337-407): where the engine wants the carrier in the IQ stream.
   * AudioSDR is not in the tree, but it is in the reference's firmware image, and asked there (its constructor and
   * setDemodMode run under tests/golden/thumb_emu.py; tests/golden/firmware_kat.npz `engine_tuning_offset`) it answers
   * as a low-IF receiver: IF centre 6890 Hz, SSB band 3000 Hz, CW band 1000 Hz, the carrier at the centre plus (lower
   * side band) or minus (upper side band) half the band; AM / SAM at the centre.  (Until round 5: 700 Hz for the CW
   * modes and 0 otherwise, build-defined.)  The engine also oscillates at this frequency itself; here the mixer is a
   * setting of its own (rdsp_*_setTuningOffsetHz), so a host that mirrors the sketch hands the value on. */
  switch (demod) {
    case RDSP_DEMOD_LSB: return 8390u;
    case RDSP_DEMOD_USB: return 5390u;
    case RDSP_DEMOD_CW_LSB: return 7390u;
    case RDSP_DEMOD_CW_USB: return 6390u;
    case RDSP_DEMOD_AM:
    case RDSP_DEMOD_SAM: return 6890u;
    default: return 0u; /* RDSP_DEMOD_IQ: the literal CONV stage, no engine in front */
  }
}

/* ---- receiver groups: double-buffered masks, records rewritten in stream order ---- */
static void group_design(rdsp_chain_t *c, GroupState &g) { /* CONV:209-224 without the upload */
  const double fs_out = c->cfg.fs_in / (double)c->decim;
  rdsp_calc_cplx_FIR_coeffs(g.coef_I.data(), g.coef_Q.data(), c->hop + 1, g.lo, g.hi, fs_out, c->cfg.window);
}

static void group_free(GroupState &g) {
  for (int i = 0; i < 2; i++) {
    if (g.ev_copy[i]) (void)hipEventDestroy(g.ev_copy[i]);
    if (g.staging[i]) (void)hipHostFree(g.staging[i]);
    g.ev_copy[i] = nullptr;
    g.staging[i] = nullptr;
    g.issued[i] = false;
  }
}

/* (re)allocate the device side for n groups; existing groups keep their settings,
 * new ones copy group 0.  Synchronous: called at create time and from
 * rdsp_chain_set_groups, never on the streaming path. */
static int groups_resize(rdsp_chain_t *c, int n) {
  HIP_TRY(hipDeviceSynchronize());
  if (!c->s_copy) {
    HIP_TRY(hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fence, hipEventDisableTiming));
  }
  const size_t old = c->groups.size();
  for (size_t i = (size_t)n; i < old; i++) group_free(c->groups[i]);
  c->groups.resize((size_t)n);
  for (size_t i = 0; i < (size_t)n; i++) {
    GroupState &g = c->groups[i];
    if (i >= old) {
      if (i > 0) {
        const GroupState &g0 = c->groups[0];
        g.lo = g0.lo; g.hi = g0.hi; g.nco_hz = g0.nco_hz; g.demod = g0.demod; g.audio_filter = g0.audio_filter;
        g.coef_I = g0.coef_I; g.coef_Q = g0.coef_Q; g.mask_nat = g0.mask_nat;
        memcpy(g.iir, g0.iir, sizeof(g.iir));
      } else {
        for (int st = 0; st < 4; st++) { g.iir[5 * st] = 1.0f; g.iir[5 * st + 1] = g.iir[5 * st + 2] = g.iir[5 * st + 3] = g.iir[5 * st + 4] = 0.0f; }
        g.coef_I.assign(c->hop + 1, 0.0);
        g.coef_Q.assign(c->hop + 1, 0.0);
        g.mask_nat.assign(2 * (size_t)c->N, 0.0f);
      }
    }
    for (int k = 0; k < 2; k++) {
      if (!g.ev_copy[k]) HIP_TRY(hipEventCreateWithFlags(&g.ev_copy[k], hipEventDisableTiming));
      if (!g.staging[k]) HIP_TRY(hipHostMalloc((void **)&g.staging[k], sizeof(float2) * (size_t)c->N, hipHostMallocDefault));
    }
  }
  if (c->d_groups) (void)hipFree(c->d_groups);
  if (c->d_mask_pool) (void)hipFree(c->d_mask_pool);
  c->d_groups = nullptr;
  c->d_mask_pool = nullptr;
  HIP_TRY(hipMalloc((void **)&c->d_groups, sizeof(RdspGroup) * (size_t)n));
  HIP_TRY(hipMemset(c->d_groups, 0, sizeof(RdspGroup) * (size_t)n));
  HIP_TRY(hipMalloc((void **)&c->d_mask_pool, sizeof(float2) * 2 * (size_t)c->N * (size_t)n));
  HIP_TRY(hipMemset(c->d_mask_pool, 0, sizeof(float2) * 2 * (size_t)c->N * (size_t)n));
  for (auto &g : c->groups) { /* pool contents are gone: every group restages */
    g.applied = 0;
    g.staged = -1;
    g.dirty = true;
  }
  c->fence_valid = false;
  return RDSP_OK;
}

/* queue the upload of group gi's current mask (all-pass when the filter is off)
 * into the buffer its device record does not point to; never blocks the
 * processing stream (the reference does this under AudioNoInterrupts, CONV:211-222) */
static int group_stage(rdsp_chain_t *c, int gi) {
  GroupState &g = c->groups[(size_t)gi];
  const int si = g.stage_next;
  /* the upload that last read this image is two retunes old: almost always long done; if not
   * (a burst of retunes of one group while the device is calls behind) the host waits for it */
  if (g.issued[si]) HIP_TRY(hipEventSynchronize(g.ev_copy[si]));
  rdsp_mask_device_image(c->cfg.filter_on ? g.mask_nat.data() : nullptr, c->N, g.staging[si]);
  const size_t img = (size_t)c->N;
  const int target = (g.staged >= 0) ? g.staged : (1 - g.applied);
  /* front kernels launched so far may still read `target` (it was live before the last switch) */
  if (c->fence_valid) HIP_TRY(hipStreamWaitEvent(c->s_copy, c->ev_fence, 0));
  float2 *dst = c->d_mask_pool + ((size_t)gi * 2 + (size_t)target) * (size_t)c->N;
  HIP_TRY(hipMemcpyAsync(dst, g.staging[si], sizeof(float2) * img, hipMemcpyHostToDevice, c->s_copy));
  HIP_TRY(hipEventRecord(g.ev_copy[si], c->s_copy));
  g.issued[si] = true;
  g.stage_last = si;
  g.stage_next = si ^ 1;
  g.staged = target;
  g.dirty = true;
  return RDSP_OK;
}

static void group_record(const rdsp_chain_t *c, const GroupState &g, int gi, int buf, RdspGroup *r) {
  memset(r, 0, sizeof(*r));
  r->dphi = rdsp_nco_dphi(g.nco_hz, c->cfg.fs_in);
  r->demod = (g.demod == RDSP_DEMOD_IQ) ? RDSP_K_DEMOD_IQ
             : (g.demod == RDSP_DEMOD_AM ? RDSP_K_DEMOD_AM
                : (g.demod == RDSP_DEMOD_SAM ? RDSP_K_DEMOD_SAM : RDSP_K_DEMOD_REAL));
  float t[2];
  rdsp_nco_rot(r->dphi, 1, t); r->rot1 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi, 2, t); r->rot2 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi, 3, t); r->rot3 = make_float2(t[0], t[1]);
  const int nt = c->N / rdsp_plan_radix(c->N); /* threads per channel */
  rdsp_nco_rot(r->dphi, 4 * nt, t); r->rotp1 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi, 8 * nt, t); r->rotp2 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi, 12 * nt, t); r->rotp3 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi, 256, t); r->rotq1 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi, 512, t); r->rotq2 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi, 768, t); r->rotq3 = make_float2(t[0], t[1]);
  r->mask_off = (uint32_t)(((size_t)gi * 2 + (size_t)buf) * (size_t)c->N);
  /* the FIR history was mixed with the increment of the launch that brought it in */
  r->dphi_hist = g.has_dev_dphi ? g.dev_dphi : r->dphi;
  rdsp_nco_rot(r->dphi_hist, 1, t); r->roth1 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi_hist, 2, t); r->roth2 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi_hist, 3, t); r->roth3 = make_float2(t[0], t[1]);
  rdsp_nco_rot(r->dphi_hist, 12 * nt, t); r->rothp3 = make_float2(t[0], t[1]);
}

/* before a launch on `stream`: switch every changed group over, in stream order */
static int groups_commit(rdsp_chain_t *c, hipStream_t stream) {
  for (size_t i = 0; i < c->groups.size(); i++) {
    GroupState &g = c->groups[i];
    if (!g.dirty) continue;
    int buf = g.applied;
    if (g.staged >= 0) {
      HIP_TRY(hipStreamWaitEvent(stream, g.ev_copy[g.stage_last], 0));
      buf = g.staged;
    }
    RdspGroup r;
    group_record(c, g, (int)i, buf, &r);
    int e = rdsp_launch_group_store(c->d_groups + i, &r, stream);
    if (e != 0) {
      rdsp_set_error("group record update failed: %s", hipGetErrorString((hipError_t)e));
      return RDSP_ERR_HIP;
    }
    g.applied = buf;
    g.staged = -1;
    /* after a tuning change the record is written once more, for the launch after this one */
    g.dirty = (r.dphi_hist != r.dphi);
    g.has_dev_dphi = true;
    g.dev_dphi = r.dphi;
  }
  return RDSP_OK;
}

static int check_group(const rdsp_chain_t *c, int group) {
  if (!c || group < 0 || (size_t)group >= c->groups.size()) {
    rdsp_set_error("group %d out of range", group);
    return RDSP_ERR_INVALID;
  }
  return RDSP_OK;
}

extern "C" int rdsp_chain_create(const rdsp_chain_config_t *cfg, int n_channels, int device,
                                 int max_blocks_per_call, rdsp_chain_t **out) {
  if (!cfg || !out || n_channels <= 0 || max_blocks_per_call <= 0) {
    rdsp_set_error("rdsp_chain_create: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (rdsp_plan_radix(cfg->fft_l) == 0) {
    rdsp_set_error("fft_l %d not in {256,512,1024,2048,4096}", cfg->fft_l);
    return RDSP_ERR_INVALID;
  }
  const int decim = cfg->decim <= 1 ? 1 : cfg->decim;
  if (decim != 1 && decim != 4) {
    rdsp_set_error("decim %d not supported (1 or 4)", cfg->decim);
    return RDSP_ERR_INVALID;
  }
  if (decim == 4 && cfg->fir_taps != 256) {
    rdsp_set_error("decimator needs 256 taps (got %d)", cfg->fir_taps);
    return RDSP_ERR_INVALID;
  }
  if (rdsp_device_count() <= 0) {
    rdsp_set_error("no HIP device: the rdsp product path has no CPU fallback");
    return RDSP_ERR_NO_DEVICE;
  }
  rdsp_chain_t *c = new rdsp_chain();
  const int rc_build = chain_build(c, cfg, n_channels, device, max_blocks_per_call, decim);
  if (rc_build != RDSP_OK) {
    rdsp_chain_destroy(c); /* frees whatever was allocated before the failure */
    return rc_build;
  }
  *out = c;
  return RDSP_OK;
}

static int chain_build(rdsp_chain_t *c, const rdsp_chain_config_t *cfg, int n_channels, int device,
                       int max_blocks_per_call, int decim) {
  c->cfg = *cfg;
  c->cfg.decim = decim;
  c->n_channels = n_channels;
  c->device = device;
  c->max_blocks = max_blocks_per_call;
  c->N = cfg->fft_l;
  c->decim = decim;
  c->hop = c->N / 2;
  c->n_in = 0;
  c->old_nr_level = 15;            /* CONV:80 */
  c->nr_mu = rdsp_lms_mu(15);      /* Init_LMS_NR(15), INO:172 */
  c->als_mu = rdsp_lms_mu(cfg->als_strength > 0 ? cfg->als_strength : 15);
  c->nr_calls = c->als_calls = 0;
  c->fir_nat.assign(256, 0.0f);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  {
    int rc = groups_resize(c, 1);
    if (rc != RDSP_OK) return rc;
    GroupState &g0 = c->groups[0];
    g0.lo = cfg->flo_hz; g0.hi = cfg->fhi_hz; g0.nco_hz = cfg->nco_hz; g0.demod = cfg->demod;
  }

  const size_t nch = (size_t)n_channels;
#define ALLOC_ZERO(ptr, bytes)                                        \
  do {                                                                \
    HIP_TRY(hipMalloc((void **)&(ptr), (bytes)));                     \
    HIP_TRY(hipMemset((ptr), 0, (bytes)));                            \
  } while (0)
  ALLOC_ZERO(c->d_fir_hc, sizeof(float) * 256);
  ALLOC_ZERO(c->d_hist, sizeof(uint32_t) * 256 * nch);
  ALLOC_ZERO(c->d_prev, sizeof(float2) * c->hop * nch);
  ALLOC_ZERO(c->d_scal, sizeof(float) * 4 * nch);
  ALLOC_ZERO(c->d_nr_w, sizeof(float) * RDSP_LMS_TAPS * nch);
  ALLOC_ZERO(c->d_nr_prev, sizeof(float) * RDSP_BLOCK * nch);
  ALLOC_ZERO(c->d_nr_energy, sizeof(float) * nch);
  ALLOC_ZERO(c->d_als_w, sizeof(float) * RDSP_LMS_TAPS * nch);
  ALLOC_ZERO(c->d_als_prev, sizeof(float) * RDSP_BLOCK * nch);
  ALLOC_ZERO(c->d_als_energy, sizeof(float) * nch);
  ALLOC_ZERO(c->d_status, sizeof(uint32_t) * 2 * nch);
  c->mid_stride = (size_t)max_blocks_per_call * RDSP_BLOCK / decim;
  ALLOC_ZERO(c->d_mid, sizeof(float) * c->mid_stride * nch);
#undef ALLOC_ZERO
  {
    std::vector<float> ones(4 * nch, 0.0f);
    for (size_t i = 0; i < nch; i++) ones[4 * i + 1] = 1.0f; /* AGC gain starts at 1 */
    HIP_TRY(hipMemcpy(c->d_scal, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (decim == 4) {
    std::vector<float> hc(256);
    if (rdsp_design_decimator(256, cfg->fir_cut_hz, cfg->fs_in, cfg->window, c->fir_nat.data(),
                              hc.data()) != 0) {
      rdsp_set_error("decimator design failed");
      return RDSP_ERR_INVALID;
    }
    HIP_TRY(hipMemcpy(c->d_fir_hc, hc.data(), sizeof(float) * 256, hipMemcpyHostToDevice));
    {
      std::vector<float> img(2 * 4 * (size_t)RDSP_FD_N);
      if (rdsp_fd_decimator_image(c->fir_nat.data(), RDSP_FD_N, img.data()) != 0) {
        rdsp_set_error("decimator spectra failed");
        return RDSP_ERR_INVALID;
      }
      HIP_TRY(hipMalloc((void **)&c->d_fd_mask, img.size() * sizeof(float)));
      HIP_TRY(hipMemcpy(c->d_fd_mask, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice));
      std::vector<float> rimg(2 * 4 * 256);
      rdsp_rd_decimator_image(c->fir_nat.data(), rimg.data());
      HIP_TRY(hipMalloc((void **)&c->d_rd_mask, rimg.size() * sizeof(float)));
      HIP_TRY(hipMemcpy(c->d_rd_mask, rimg.data(), rimg.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  /* boot order of the sketch: doConvolutionalInitialize (INO:180, mask from the
   * still-zero taps) then reInitializeFilter (INO:183) */
  int rc = rdsp_doConvolutionalInitialize(c, nullptr);
  if (rc == RDSP_OK) rc = rdsp_reInitializeFilter(c, cfg->flo_hz, cfg->fhi_hz, nullptr);
  if (rc == RDSP_OK && cfg->demod == RDSP_DEMOD_SAM) rc = ensure_sam(c);
  return rc;
}

extern "C" void rdsp_chain_destroy(rdsp_chain_t *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->engine) rdsp_engine_destroy(c->engine);
  if (c->pre) rdsp_preproc_destroy(c->pre);
  if (c->d_engine_io) (void)hipFree(c->d_engine_io);
  void *ptrs[] = {c->d_iir_coef, c->d_iir_state, c->d_fd_mask, c->d_rd_mask, c->d_sin_table, c->d_mid_q[0], c->d_mid_q[1], c->d_mid_q[2], c->d_sam, c->d_groups, c->d_group_of, c->d_mask_pool, c->d_fir_hc, c->d_hist, c->d_prev, c->d_scal,
                  c->d_nr_w, c->d_nr_prev, c->d_nr_energy, c->d_als_w, c->d_als_prev,
                  c->d_als_energy, c->d_status, c->d_mid, c->d_slip_buf, c->d_slip_carry[0], c->d_slip_carry[1]};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  if (c->s_copy) {
    (void)hipStreamSynchronize(c->s_copy);
    (void)hipStreamDestroy(c->s_copy);
    (void)hipEventDestroy(c->ev_fence);
  }
  for (auto &g : c->groups) group_free(g);
  if (c->s_tail) {
    if (c->s_mid) {
      (void)hipStreamSynchronize(c->s_mid);
      (void)hipStreamDestroy(c->s_mid);
      for (int i = 0; i < 3; i++) (void)hipEventDestroy(c->ev_mid[i]);
    }
    (void)hipStreamSynchronize(c->s_tail);
    (void)hipStreamDestroy(c->s_tail);
    for (int i = 0; i < 3; i++) { (void)hipEventDestroy(c->ev_front[i]); (void)hipEventDestroy(c->ev_tail[i]); }
    for (int i = 0; i < 3; i++) for (hipEvent_t ev : c->ev_front_sb[i]) (void)hipEventDestroy(ev);
    (void)hipEventDestroy(c->ev_misc);
    (void)hipFree(c->d_midx[0]);
    (void)hipFree(c->d_midx[1]);
  }
  delete c;
}

extern "C" int rdsp_chain_channels(const rdsp_chain_t *c) { return c ? c->n_channels : 0; }
extern "C" int rdsp_chain_decim(const rdsp_chain_t *c) { return c ? c->decim : 0; }
extern "C" int rdsp_chain_device(const rdsp_chain_t *c) { return c ? c->device : -1; }

/* RdspFrontParams::fir_fd of a chain: 0 direct form (and every decim-1 chain), 1 / 2 the wave-wide frequency-domain
 * forms (448-sample frames / one granule per frame), 3 / 4 the row forms (128 / 192 outputs per window) */
static int fir_fd_of(const rdsp_chain_t *c) {
  if (!c->d_fd_mask) return 0;
  switch (c->fir_mode) {
    case 2: return 1;
    case -1: case 4: return 2;
    case 5: return 3;
    case 6: return 4;
    default: return 0;
  }
}
/* the smallest call: one kernel chunk = 256 output samples; an overlap-save frame needs fft_l/2 of them */
extern "C" int rdsp_chain_call_unit_blocks(const rdsp_chain_t *c) {
  if (!c) return 0;
  const int out_samples = c->hop > 256 ? c->hop : 256;
  return out_samples * c->decim / RDSP_BLOCK;
}
/* the unit a stream is cut in for bits that do not depend on the cut.  Default decimator, direct form, decim 1:
 * the call unit.  448-sample decimator frames (fir_variant 2; 14 input blocks each): the least common multiple of
 * a frame and the call unit -- calls of whole granules are whole frames, so the frame grid sits at absolute stream
 * positions whatever the split (the reference restricts its call boundaries the same way: `available() >
 * N_BLOCKS`, CONV:231) */
extern "C" int rdsp_chain_granule_blocks(const rdsp_chain_t *c) {
  if (!c) return 0;
  const int unit = rdsp_chain_call_unit_blocks(c);
  if ((c->fir_mode != 2 && c->fir_mode != 6) || !c->d_fd_mask) return unit;
  const int frame = c->fir_mode == 2 ? 14 : 6; /* 448 (192) outputs x 4 / 128 */
  int a = unit, b = frame;
  while (b) { const int t = a % b; a = b; b = t; }
  return unit / a * frame;
}

extern "C" int rdsp_chain_reset(rdsp_chain_t *c, void *stream_) {
  if (!c) return RDSP_ERR_INVALID;
  hipStream_t stream = (hipStream_t)stream_;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  const size_t nch = (size_t)c->n_channels;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  c->call_idx = 0;
  c->tail_slot = -1;
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(hipMemset(c->d_hist, 0, sizeof(uint32_t) * 256 * nch));
  HIP_TRY(hipMemset(c->d_prev, 0, sizeof(float2) * c->hop * nch));
  HIP_TRY(hipMemset(c->d_nr_w, 0, sizeof(float) * RDSP_LMS_TAPS * nch));
  HIP_TRY(hipMemset(c->d_nr_prev, 0, sizeof(float) * RDSP_BLOCK * nch));
  HIP_TRY(hipMemset(c->d_nr_energy, 0, sizeof(float) * nch));
  HIP_TRY(hipMemset(c->d_als_w, 0, sizeof(float) * RDSP_LMS_TAPS * nch));
  HIP_TRY(hipMemset(c->d_als_prev, 0, sizeof(float) * RDSP_BLOCK * nch));
  HIP_TRY(hipMemset(c->d_als_energy, 0, sizeof(float) * nch));
  HIP_TRY(hipMemset(c->d_status, 0, sizeof(uint32_t) * 2 * nch));
  c->slip_prev_on = false;
  std::vector<float> sc(4 * nch, 0.0f);
  for (size_t i = 0; i < nch; i++) sc[4 * i + 1] = 1.0f;
  HIP_TRY(hipMemcpy(c->d_scal, sc.data(), sc.size() * sizeof(float), hipMemcpyHostToDevice));
  c->n_in = 0;
  c->hist_valid = false;
  c->nr_calls = c->als_calls = 0;
  c->old_nr_level = 15;
  c->nr_mu = rdsp_lms_mu(15);
  for (auto &g : c->groups) { g.has_dev_dphi = false; g.dirty = true; }
  if (c->d_sam) HIP_TRY(hipMemset(c->d_sam, 0, sizeof(float) * 4 * nch));
  if (c->d_iir_state) HIP_TRY(hipMemset(c->d_iir_state, 0, sizeof(float) * 16 * nch));
  return RDSP_OK;
}

/* CONV:187-207 for one group: build the mask from whatever the tap arrays hold */
static int group_initialize(rdsp_chain_t *c, int gi) {
  GroupState &g = c->groups[(size_t)gi];
  if (rdsp_init_filter_mask(g.mask_nat.data(), g.coef_I.data(), g.coef_Q.data(), c->N) != 0) {
    rdsp_set_error("init_filter_mask failed");
    return RDSP_ERR_INVALID;
  }
  return group_stage(c, gi);
}

extern "C" int rdsp_doConvolutionalInitialize(rdsp_chain_t *c, void *stream) {
  (void)stream; /* the new mask is switched in by the next processing call, in its stream's order */
  if (!c) return RDSP_ERR_INVALID;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  for (size_t i = 0; i < c->groups.size(); i++) {
    int rc = group_initialize(c, (int)i);
    if (rc != RDSP_OK) return rc;
  }
  return RDSP_OK;
}

/* CONV:209-224 for one group.  Host-side design, asynchronous upload into
 * the group's idle mask buffer; processing never waits on the host. */
extern "C" int rdsp_group_reInitializeFilter(rdsp_chain_t *c, int group, double lo, double hi, void *stream) {
  (void)stream;
  if (check_group(c, group) != RDSP_OK) return RDSP_ERR_INVALID;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  GroupState &g = c->groups[(size_t)group];
  g.lo = lo;
  g.hi = hi;
  group_design(c, g);
  if (group == 0) { c->cfg.flo_hz = lo; c->cfg.fhi_hz = hi; }
  return group_initialize(c, group);
}

/* CONV:209-224: every group gets the same band */
extern "C" int rdsp_reInitializeFilter(rdsp_chain_t *c, double lo, double hi, void *stream) {
  if (!c) return RDSP_ERR_INVALID;
  for (size_t i = 0; i < c->groups.size(); i++) {
    int rc = rdsp_group_reInitializeFilter(c, (int)i, lo, hi, stream);
    if (rc != RDSP_OK) return rc;
  }
  return RDSP_OK;
}

/* NR:35-64: new mu; delay line and filter state cleared, energy = 0; the
 * coefficients are NOT cleared (arm_lms_norm_init_f32 leaves them) */
extern "C" int rdsp_Init_LMS_NR(rdsp_chain_t *c, int strength, void *stream_) {
  if (!c) return RDSP_ERR_INVALID;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  hipStream_t stream = (hipStream_t)stream_;
  c->nr_mu = rdsp_lms_mu(strength);
  const size_t nch = (size_t)c->n_channels;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail)); /* the tail stage owns these arrays */
  HIP_TRY(hipMemsetAsync(c->d_nr_prev, 0, sizeof(float) * RDSP_BLOCK * nch, stream));
  HIP_TRY(hipMemsetAsync(c->d_nr_energy, 0, sizeof(float) * nch, stream));
  HIP_TRY(hipMemsetAsync(c->d_status, 0, sizeof(uint32_t) * nch, stream)); /* the DSP-NR instance's health words */
  if (c->s_tail) {
    HIP_TRY(hipEventRecord(c->ev_misc, stream));
    HIP_TRY(hipStreamWaitEvent(c->s_tail, c->ev_misc, 0));
  }
  return RDSP_OK;
}

extern "C" int rdsp_chain_process(rdsp_chain_t *c, const int16_t *d_iq, size_t in_stride,
                                  int n_blocks, int16_t *d_out, size_t out_stride,
                                  float *d_out_f32, void *stream_) {
  if (!c || !d_iq || !d_out || n_blocks <= 0) {
    rdsp_set_error("rdsp_chain_process: bad argument");
    return RDSP_ERR_INVALID;
  }
  const int gran = rdsp_chain_call_unit_blocks(c);
  if (n_blocks % gran != 0) {
    rdsp_set_error("n_blocks %d is not a multiple of the call unit %d", n_blocks, gran);
    return RDSP_ERR_NOT_READY;
  }
  if (n_blocks > c->max_blocks) {
    rdsp_set_error("n_blocks %d exceeds max_blocks_per_call %d", n_blocks, c->max_blocks);
    return RDSP_ERR_INVALID;
  }
  const size_t n_in = (size_t)n_blocks * RDSP_BLOCK;
  const size_t n_out = n_in / c->decim;
  if (in_stride < n_in || out_stride < n_out || (in_stride & 3) != 0 ||
      ((uintptr_t)d_iq & 15) != 0 || ((uintptr_t)d_out & 15) != 0 || (out_stride & 3) != 0) {
    rdsp_set_error("strides/alignment: in_stride %zu (>= %zu, %%4), out_stride %zu (>= %zu, %%4), 16-byte aligned bases",
                   in_stride, n_in, out_stride, n_out);
    return RDSP_ERR_INVALID;
  }
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  hipStream_t stream = (hipStream_t)stream_;
  const rdsp_chain_config_t &cf = c->cfg;
  if (c->engine) { /* INO:71-72,81-86: IQ -> preProcessor -> SDR -> the record queues; this chain is the CONV stage behind them */
    const size_t st = (size_t)c->max_blocks * RDSP_BLOCK;
    int rc = rdsp_preproc_update(c->pre, d_iq, in_stride, n_blocks, c->d_engine_io, st, stream);
    if (rc == RDSP_OK) rc = rdsp_engine_update(c->engine, c->d_engine_io, st, n_blocks, c->d_engine_io, st, stream);
    if (rc != RDSP_OK) return rc;
    d_iq = c->d_engine_io;
    in_stride = st;
  }

  /* CONV:326-330: nr level change re-initialises the NLMS instance */
  if (cf.lms_nr > 0 && cf.lms_nr != c->old_nr_level) {
    int rc = rdsp_Init_LMS_NR(c, cf.lms_nr, stream);
    if (rc != RDSP_OK) return rc;
    c->old_nr_level = cf.lms_nr;
  }
  bool sam = false; /* any group on the PLL demodulator: its serial stage runs before the tail */
  for (const auto &g : c->groups) sam = sam || (g.demod == RDSP_DEMOD_SAM);
  if (sam && !c->d_sam) { /* rdsp_*_setDemodMode(SAM) allocates them; nothing is allocated here */
    rdsp_set_error("SAM group without PLL buffers (internal)");
    return RDSP_ERR_INVALID;
  }
  const bool iir = c->audio_kind == RDSP_AUDIO_KIND_IIR && cf.filter_on;
  if (iir && (!c->d_iir_coef || c->iir_sets < (int)c->groups.size())) {
    rdsp_set_error("IIR audio filter without its buffers (internal)");
    return RDSP_ERR_INVALID;
  }
  const bool tail = sam || iir || (cf.lms_nr > 0) || (cf.als_mode != RDSP_ALS_OFF);
  float attack, decay;
  agc_params(cf.agc_mode, &attack, &decay);
  const float og = cf.mute ? 0.0f : cf.output_gain;

  RdspFrontParams fp;
  memset(&fp, 0, sizeof(fp));
  fp.iq = reinterpret_cast<const uint32_t *>(d_iq);
  fp.in_stride = in_stride;
  if (c->iq_slip != 0) {
    /* the word in front of this call's first sample: the previous pass's carry, or -- when the previous
     * call ran without the correction -- the last word of the FIR history, which is raw then */
    const size_t sstride = (size_t)c->max_blocks * RDSP_BLOCK;
    const uint32_t *cin = c->slip_prev_on ? c->d_slip_carry[c->slip_phase] : c->d_hist + 255;
    int e = rdsp_launch_iq_slip(fp.iq, in_stride, c->d_slip_buf, sstride, cin, c->slip_prev_on ? 1 : 256,
                                c->d_slip_carry[c->slip_phase ^ 1], (int)n_in, c->iq_slip, c->n_channels, stream);
    if (e != 0) {
      rdsp_set_error("slip kernel launch failed: %s", hipGetErrorString((hipError_t)e));
      return RDSP_ERR_HIP;
    }
    fp.iq = c->d_slip_buf;
    fp.in_stride = sstride;
    c->slip_phase ^= 1;
  }
  c->slip_prev_on = c->iq_slip != 0;
  fp.n_chunks = (int)(n_in / (size_t)(256 * c->decim));
  fp.n0 = (uint32_t)c->n_in;
  fp.scale_i = cf.iq_balance * cf.input_gain * (1.0f / 32768.0f);
  fp.scale_q = cf.input_gain * (1.0f / 32768.0f);
  fp.swap_iq = c->swap_iq;
  fp.swap_hist = c->hist_valid ? c->hist_swap : fp.swap_iq;
  fp.scale_i_hist = c->hist_valid ? c->hist_scale_i : fp.scale_i;
  fp.scale_q_hist = c->hist_valid ? c->hist_scale_q : fp.scale_q;
  fp.nb_on = c->nb_on;
  fp.nb_thr = (float)pow(10.0, (double)c->nb_threshold_db / 10.0);
  fp.fir_hc = c->d_fir_hc;
  fp.groups = c->d_groups;
  fp.group_of = c->d_group_of;
  fp.mask_pool = c->d_mask_pool;
  fp.spectral_on = cf.spectral_nr == 2 ? 2 : (cf.spectral_nr ? 1 : 0);
  if (cf.spectral_nr == 2) { /* older variant, backup/RadioDSP_SDR_RX_Conv.ino:1594-1596: bins 60..120, x3 */
    fp.spectral_k = 3.0f;
    fp.vad_lo = 60 * c->N / 256;
    fp.vad_hi = 120 * c->N / 256;
  } else {
    fp.spectral_k = (float)((double)cf.spectral_level * 1.5);
    fp.vad_lo = 30 * c->N / 256; /* STATING_BIN_VAD_ANALISYS, SPEC:34, scaled with FFT_L */
    fp.vad_hi = 180 * c->N / 256;
  }
  fp.spectral_literal = (c->spectral_literal && cf.spectral_nr == 1) ? c->spectral_literal : 0; /* SPEC only: the older variant scales the bin (BK_INO:1614-1628) */
  fp.sin_table = c->d_sin_table;
  fp.to_mid = tail ? 1 : 0;
  fp.agc_on = cf.agc_mode != RDSP_AGC_OFF;
  fp.agc_attack = attack;
  fp.agc_decay = decay;
  fp.out_gain = og;
  fp.st_hist = c->d_hist;
  fp.st_prev = c->d_prev;
  fp.st_scal = c->d_scal;
  fp.out_i16 = reinterpret_cast<uint32_t *>(d_out);
  fp.out_stride = out_stride;
  fp.out_f32 = reinterpret_cast<float2 *>(d_out_f32);
  fp.mid = c->d_mid;
  fp.mid_stride = c->mid_stride;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
  const bool timed = c->timing_on && 4 * (c->ev_used + 1) <= c->ev.size();
  const bool piped = tail && c->pipe_on;
  const int slot = (int)(c->call_idx % 3);
  hipStream_t tstream = piped ? c->s_tail : stream;
  const bool mid_stage = sam || iir;
  hipStream_t mstream = (piped && mid_stage) ? c->s_mid : tstream; /* SAM PLL / IIR cascade */
  /* full-register front kernel in both modes: since its butterflies shrank to 199 VGPRs two of
   * its waves and a tail wave fit one SIMD, and the lean variant's twiddle chains only cost */
  fp.lean = (c->lean_mode < 0) ? 0 : c->lean_mode;
  fp.front_prio = piped ? c->front_fir_prio : 0;
  fp.fir_matrix = (c->fir_mode == 3) ? (piped ? 0 : 1) : (c->fir_mode == 1);
  /* stage A3 (rdsp_chain_set_fir_variant).  Default (-1) and 4: in the frequency domain with frames of one granule
   * (256 outputs per 512-point window): every frame's input is a function of the absolute sample position, so the
   * bits do not depend on how the stream is cut into calls -- like the direct form (0), at about two thirds of
   * its cost.  2: 448-sample frames anchored at the call's first sample: the throughput form bench.py selects. */
  fp.fir_fd = fir_fd_of(c);
  /* the default (-1) picks between the two split-invariant frequency-domain forms: on 16-lane rows where no tail kernel
   * will share the SIMDs with this call's front kernel (4-10 % faster there), wave-wide frames of one granule beside
   * it (the rows' 233-256 registers would leave one front wave per SIMD).  A function of the chain's settings at this
   * call, not of how the stream is cut: both give the same bits for any split. */
  if (c->fir_mode == -1 && fp.fir_fd == 2 && !fp.to_mid) fp.fir_fd = 3;
  fp.fd_mask = c->d_fd_mask;
  fp.rd_mask = c->d_rd_mask;
  c->front_name = !fp.fir_fd ? "rdsp_front_kernel" : ((fp.fir_fd >= 3 && !fp.nb_on) ? "rdsp_front_rd_kernel" : "rdsp_front_fd_kernel");
  fp.mid_q = c->d_mid_q[0];
  if (piped) {
    fp.mid = slot ? c->d_midx[slot - 1] : c->d_mid;
    fp.mid_q = c->d_mid_q[slot];
    /* the tail of call k-2 read this intermediate buffer: wait for it */
    if (c->call_idx >= 3) HIP_TRY(hipStreamWaitEvent(stream, c->ev_tail[slot], 0));
  }
  if (!piped && c->tail_slot >= 0) {
    /* the previous call's tail stage may still be running on s_tail: it owns that call's d_out and
     * updates the AGC gain (st_scal) this call's front kernel reads and writes when it packs itself
     * (tail stage switched off between two calls), and the NLMS state an in-stream tail uses */
    HIP_TRY(hipStreamWaitEvent(stream, c->ev_tail[c->tail_slot], 0));
    c->tail_slot = -1;
  }
  if (timed) { /* events come from a pool created in rdsp_chain_set_timing */
    ev0 = c->ev[4 * c->ev_used];
    ev1 = c->ev[4 * c->ev_used + 1];
    ev2 = c->ev[4 * c->ev_used + 2];
    ev3 = c->ev[4 * c->ev_used + 3];
    HIP_TRY(hipEventRecord(ev0, stream));
  }
  {
    /* the PLL kernel of the previous call (on s_tail) reads the group records: a record is only
     * rewritten after it has finished */
    bool any_dirty = false;
    for (const auto &g : c->groups) any_dirty = any_dirty || g.dirty;
    if (any_dirty && c->d_sam && c->tail_slot >= 0) HIP_TRY(hipStreamWaitEvent(stream, c->ev_tail[c->tail_slot], 0));
    int rc = groups_commit(c, stream);
    if (rc != RDSP_OK) return rc;
  }
  /* channel sub-batches (see sub_batch): only where the tail runs on its own stream */
  int nsb = 1, sbn = c->n_channels;
  if (piped && !sam && c->sub_batch > 0 && c->n_channels >= c->sub_batch + c->sub_batch / 2) {
    sbn = c->sub_batch;
    nsb = (c->n_channels + sbn - 1) / sbn;
    if (c->ev_front_sb[slot].size() < (size_t)nsb) { /* made by set_pipelined / set_sub_batch */
      rdsp_set_error("sub-batch events missing (internal)");
      return RDSP_ERR_INVALID;
    }
  }
  int e = 0;
  for (int k = 0; k < nsb && e == 0; k++) {
    fp.ch_base = k * sbn;
    const int count = (c->n_channels - fp.ch_base < sbn) ? c->n_channels - fp.ch_base : sbn;
    e = rdsp_launch_front(c->N, c->decim, &fp, count, stream);
    if (nsb > 1 && e == 0) HIP_TRY(hipEventRecord(c->ev_front_sb[slot][k], stream));
  }
  if (timed) HIP_TRY(hipEventRecord(ev1, stream));
  HIP_TRY(hipEventRecord(c->ev_fence, stream));
  c->fence_valid = true;
  if (e != 0) {
    rdsp_set_error("front kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return RDSP_ERR_HIP;
  }
  if (piped && nsb == 1) {
    HIP_TRY(hipEventRecord(c->ev_front[slot], stream));
    HIP_TRY(hipStreamWaitEvent(mid_stage ? c->s_mid : c->s_tail, c->ev_front[slot], 0));
  }
  if (sam) {
    RdspSamParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.mid = fp.mid;
    sp.mid_q = fp.mid_q;
    sp.mid_stride = c->mid_stride;
    sp.n_channels = c->n_channels;
    sp.n_samples = (int)n_out;
    sp.groups = c->d_groups;
    sp.group_of = c->d_group_of;
    rdsp_sam_constants(cf.fs_in / (double)c->decim, &sp.g1, &sp.g2, &sp.wmin, &sp.wmax);
    sp.st_sam = c->d_sam;
    int es = rdsp_launch_sam(&sp, mstream);
    if (es != 0) {
      rdsp_set_error("SAM kernel launch failed: %s", hipGetErrorString((hipError_t)es));
      return RDSP_ERR_HIP;
    }
  }
  if (iir) { /* SDR.setAudioFilter as a biquad cascade on the demodulated audio, before NR / notch / AGC */
    for (size_t gi = 0; gi < c->groups.size(); gi++) {
      GroupState &g = c->groups[gi];
      if (!g.iir_dirty) continue;
      int eb = rdsp_launch_biquad_coef_store(c->d_iir_coef + 20 * gi, g.iir, mstream);
      if (eb != 0) {
        rdsp_set_error("IIR coefficient store failed: %s", hipGetErrorString((hipError_t)eb));
        return RDSP_ERR_HIP;
      }
      g.iir_dirty = false;
    }
    RdspBiquadParams bp;
    memset(&bp, 0, sizeof(bp));
    bp.buf = fp.mid;
    bp.stride = c->mid_stride;
    bp.n_channels = c->n_channels;
    bp.n_samples = (int)n_out;
    bp.coef = c->d_iir_coef;
    bp.set_of = c->d_group_of;
    bp.state = c->d_iir_state;
    if (nsb > 1) /* sub-batched fronts: the cascade covers all channels, after the last of them */
      for (int k = 0; k < nsb; k++) HIP_TRY(hipStreamWaitEvent(mstream, c->ev_front_sb[slot][k], 0));
    int eb = rdsp_launch_biquad(&bp, mstream);
    if (eb != 0) {
      rdsp_set_error("biquad kernel launch failed: %s", hipGetErrorString((hipError_t)eb));
      return RDSP_ERR_HIP;
    }
  }
  if (piped && mid_stage) { /* the tail stage of this call follows its PLL / cascade */
    HIP_TRY(hipEventRecord(c->ev_mid[slot], c->s_mid));
    HIP_TRY(hipStreamWaitEvent(c->s_tail, c->ev_mid[slot], 0));
  }
  if (tail) {
    RdspTailParams tp;
    memset(&tp, 0, sizeof(tp));
    tp.mid = fp.mid;
    tp.mid_stride = c->mid_stride;
    tp.n_channels = c->n_channels;
    tp.n_blocks = (int)(n_out / RDSP_BLOCK);
    tp.nr_on = cf.lms_nr > 0;
    tp.als_mode = cf.als_mode;
    tp.nr_mu = c->nr_mu;
    tp.als_mu = c->als_mu;
    tp.nr_first = (c->nr_calls == 0);
    tp.als_first = (c->als_calls == 0);
    tp.nr_w = c->d_nr_w; tp.nr_prev = c->d_nr_prev; tp.nr_energy = c->d_nr_energy;
    tp.als_w = c->d_als_w; tp.als_prev = c->d_als_prev; tp.als_energy = c->d_als_energy;
    tp.agc_on = fp.agc_on;
    tp.agc_attack = attack;
    tp.agc_decay = decay;
    tp.out_gain = og;
    tp.st_scal = c->d_scal;
    tp.st_status = c->d_status;
    tp.st_status_stride = (size_t)c->n_channels;
    tp.prio = piped ? c->tail_prio : 0;
    tp.energy_running = c->nlms_energy_running;
    tp.out_i16 = reinterpret_cast<uint32_t *>(d_out);
    tp.out_stride = out_stride;
    tp.out_f32 = reinterpret_cast<float2 *>(d_out_f32);
    if (timed) HIP_TRY(hipEventRecord(ev2, tstream));
    for (int k = 0; k < nsb && e == 0; k++) {
      tp.ch_base = k * sbn;
      tp.n_channels = (c->n_channels - tp.ch_base < sbn) ? c->n_channels : tp.ch_base + sbn;
      if (nsb > 1) HIP_TRY(hipStreamWaitEvent(c->s_tail, c->ev_front_sb[slot][k], 0));
      e = rdsp_launch_tail(&tp, c->tail_lpc, tstream);
    }
    if (e != 0) {
      rdsp_set_error("tail kernel launch failed: %s", hipGetErrorString((hipError_t)e));
      return RDSP_ERR_HIP;
    }
    if (timed) HIP_TRY(hipEventRecord(ev3, tstream));
    if (piped) {
      HIP_TRY(hipEventRecord(c->ev_tail[slot], c->s_tail));
      c->tail_slot = slot;
    }
    if (tp.nr_on) c->nr_calls += tp.n_blocks;
    if (tp.als_mode) c->als_calls += tp.n_blocks;
  }
  if (timed) {
    c->ev_has_tail[c->ev_used] = tail ? 1 : 0;
    c->ev_used++;
  }
  c->call_idx++;
  c->n_in += n_in;
  c->hist_valid = true;
  c->hist_swap = fp.swap_iq;
  c->hist_scale_i = fp.scale_i;
  c->hist_scale_q = fp.scale_q;
  return RDSP_OK;
}

/* LMS_NoiseReduction(blockSize, nrbuffer), NR:66-80: the DSP-NR instance alone, in
 * place on float buffers [n_channels][stride], n_samples a multiple of 128 */
extern "C" int rdsp_LMS_NoiseReduction(rdsp_chain_t *c, int n_samples, float *d_nrbuffer,
                                       size_t stride, void *stream_) {
  if (!c || !d_nrbuffer || n_samples <= 0 || n_samples % RDSP_BLOCK != 0 || stride < (size_t)n_samples ||
      (stride & 3) != 0 || ((uintptr_t)d_nrbuffer & 15) != 0) {
    rdsp_set_error("rdsp_LMS_NoiseReduction: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  RdspTailParams tp;
  memset(&tp, 0, sizeof(tp));
  tp.mid = d_nrbuffer;
  tp.mid_stride = stride;
  tp.raw_out = d_nrbuffer;
  tp.n_channels = c->n_channels;
  tp.n_blocks = n_samples / RDSP_BLOCK;
  tp.nr_on = 1;
  tp.nr_mode = 2;
  tp.nr_mu = c->nr_mu;
  tp.nr_first = (c->nr_calls == 0);
  tp.energy_running = c->nlms_energy_running;
  tp.nr_w = c->d_nr_w; tp.nr_prev = c->d_nr_prev; tp.nr_energy = c->d_nr_energy;
  tp.st_scal = c->d_scal;
  tp.st_status = c->d_status;
  tp.st_status_stride = (size_t)c->n_channels;
  int e = rdsp_launch_tail(&tp, c->tail_lpc, (hipStream_t)stream_);
  if (e != 0) {
    rdsp_set_error("tail kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    return RDSP_ERR_HIP;
  }
  c->nr_calls += tp.n_blocks;
  return RDSP_OK;
}

extern "C" int rdsp_doConvolutionalProcessing(rdsp_chain_t *c, float iNRLevel, int bFilterEnabled,
                                              double dFLoCut, double dFHiCut, const int16_t *d_iq,
                                              size_t in_stride, int n_blocks, int16_t *d_out,
                                              size_t out_stride, void *stream) {
  (void)dFLoCut; /* ignored by the reference too: only bFilterEnabled is read, CONV:300 */
  (void)dFHiCut;
  if (!c) return RDSP_ERR_INVALID;
  c->cfg.lms_nr = (int)iNRLevel;
  const int f = bFilterEnabled ? 1 : 0;
  if (f != c->cfg.filter_on) {
    c->cfg.filter_on = f;
    for (size_t i = 0; i < c->groups.size(); i++) {
      int rc = group_stage(c, (int)i);
      if (rc != RDSP_OK) return rc;
    }
  }
  return rdsp_chain_process(c, d_iq, in_stride, n_blocks, d_out, out_stride, nullptr, stream);
}

extern "C" int rdsp_q15_to_float(const int16_t *d_src, float *d_dst, size_t n, void *stream) {
  if (rdsp_device_count() <= 0) { rdsp_set_error("no HIP device"); return RDSP_ERR_NO_DEVICE; }
  int e = rdsp_launch_q15_to_float(d_src, d_dst, n, (hipStream_t)stream);
  if (e) { rdsp_set_error("q15_to_float launch: %s", hipGetErrorString((hipError_t)e)); return RDSP_ERR_HIP; }
  return RDSP_OK;
}
extern "C" int rdsp_float_to_q15(const float *d_src, int16_t *d_dst, size_t n, void *stream) {
  if (rdsp_device_count() <= 0) { rdsp_set_error("no HIP device"); return RDSP_ERR_NO_DEVICE; }
  int e = rdsp_launch_float_to_q15(d_src, d_dst, n, (hipStream_t)stream);
  if (e) { rdsp_set_error("float_to_q15 launch: %s", hipGetErrorString((hipError_t)e)); return RDSP_ERR_HIP; }
  return RDSP_OK;
}

/* ---- engine setters ------------------------------------------------------- */
#define NEED(c) do { if (!(c)) return RDSP_ERR_INVALID; } while (0)
/* with rdsp_sdr_set_engine_literal(chain, 1) the `SDR.` / `preProcessor.` calls reach the reference's own objects */
#define TO_ENGINE(c, call) do { if ((c) && (c)->engine) return (call); } while (0)
static int engine_mode_of(int demod) { /* rdsp_demod_t -> the engine's numbering (as the compiled tuningMode() passes it) */
  switch (demod) {
    case RDSP_DEMOD_LSB: return 0; case RDSP_DEMOD_USB: return 1; case RDSP_DEMOD_CW_LSB: return 2; case RDSP_DEMOD_CW_USB: return 3;
    case RDSP_DEMOD_AM: return 4; case RDSP_DEMOD_SAM: return 5; default: return -1;
  }
}
static int engine_filter_of(int filter) { /* rdsp_audio_filter_t -> the engine's id (as the compiled filterMode() passes it) */
  switch (filter) {
    case RDSP_AUDIO_AM: return 0; case RDSP_AUDIO_CW: return 1; case RDSP_AUDIO_2100: return 3; case RDSP_AUDIO_2700: return 6;
    case RDSP_AUDIO_3100: return 8; default: return -1;
  }
}
extern "C" int rdsp_sdr_enableAGC(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_enableAGC(c->engine)); if (c->cfg.agc_mode == RDSP_AGC_OFF) c->cfg.agc_mode = c->saved_agc_mode; return RDSP_OK; }
extern "C" int rdsp_sdr_disableAGC(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_setAGCmode(c->engine, 0)); if (c->cfg.agc_mode != RDSP_AGC_OFF) c->saved_agc_mode = c->cfg.agc_mode; c->cfg.agc_mode = RDSP_AGC_OFF; return RDSP_OK; }
extern "C" int rdsp_sdr_setAGCmode(rdsp_chain_t *c, int mode) {
  NEED(c);
  TO_ENGINE(c, rdsp_engine_setAGCmode(c->engine, mode));
  if (mode < RDSP_AGC_OFF || mode > RDSP_AGC_SLOW) return RDSP_ERR_INVALID;
  c->cfg.agc_mode = mode;
  return RDSP_OK;
}
extern "C" int rdsp_sdr_enableALSfilter(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_enableALSfilter(c->engine)); if (c->cfg.als_mode == RDSP_ALS_OFF) c->cfg.als_mode = c->saved_als_mode; return RDSP_OK; }
extern "C" int rdsp_sdr_disableALSfilter(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_disableALSfilter(c->engine)); if (c->cfg.als_mode != RDSP_ALS_OFF) c->saved_als_mode = c->cfg.als_mode; c->cfg.als_mode = RDSP_ALS_OFF; return RDSP_OK; }
extern "C" int rdsp_sdr_setALSfilterNotch(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_setALSfilterNotch(c->engine)); c->saved_als_mode = RDSP_ALS_NOTCH; if (c->cfg.als_mode != RDSP_ALS_OFF) c->cfg.als_mode = RDSP_ALS_NOTCH; return RDSP_OK; }
extern "C" int rdsp_sdr_setALSfilterPeak(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_setALSfilterPeak(c->engine)); c->saved_als_mode = RDSP_ALS_PEAK; if (c->cfg.als_mode != RDSP_ALS_OFF) c->cfg.als_mode = RDSP_ALS_PEAK; return RDSP_OK; }
extern "C" int rdsp_sdr_setALSfilterAdaptive(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_setALSfilterAdaptive(c->engine)); return RDSP_OK; /* the NLMS always adapts */ }
/* noise blanker (AudioSDR feature; arithmetic build-defined): wide-band,
 * before the mixer; windows of 256*decim input samples */
extern "C" int rdsp_sdr_enableNoiseBlanker(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_enableNoiseBlanker(c->engine)); c->nb_on = 1; return RDSP_OK; }
extern "C" int rdsp_sdr_disableNoiseBlanker(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_engine_disableNoiseBlanker(c->engine)); c->nb_on = 0; return RDSP_OK; }
extern "C" int rdsp_sdr_setNoiseBlankerThresholdDb(rdsp_chain_t *c, float db) {
  NEED(c);
  if (!(db >= 0.0f && db <= 60.0f)) { rdsp_set_error("noise blanker threshold %g dB outside 0..60", (double)db); return RDSP_ERR_INVALID; }
  c->nb_threshold_db = db;
  return RDSP_OK;
}
/* AudioSDRpreProcessor (INO:117-118) */
extern "C" int rdsp_pre_swapIQ(rdsp_chain_t *c, int swap) { NEED(c); TO_ENGINE(c, rdsp_preproc_swapIQ(c->pre, swap)); c->swap_iq = swap ? 1 : 0; return RDSP_OK; }
/* INO:117 guards against a Teensy I2S bus fault that leaves one rail of the codec stream a sample
 * behind the other.  There is no bus here, so there is nothing to watch at run time; a RECORDING made
 * through such a front end carries the fault: rdsp_estimate_iq_slip finds it, rdsp_pre_setIQslip
 * corrects it. */
extern "C" int rdsp_pre_startAutoI2SerrorDetection(rdsp_chain_t *c) { NEED(c); TO_ENGINE(c, rdsp_preproc_startAutoI2SerrorDetection(c->pre)); return RDSP_OK; }
/* slip +1: pair I[n-1] with Q[n] (delay the I rail by one sample); -1: pair I[n] with Q[n-1]; 0: off.
 * Applies to samples as they arrive, from the next call on (what is already in the FIR history keeps
 * the pairing it came in with).  A set-up call: the first non-zero value allocates the corrected-input
 * buffer ([n_channels][max_blocks_per_call * 128] words). */
extern "C" int rdsp_pre_setIQslip(rdsp_chain_t *c, int slip) {
  NEED(c);
  if (slip < -1 || slip > 1) return RDSP_ERR_INVALID;
  if (slip != 0 && !c->d_slip_buf) {
    if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
    const size_t nch = (size_t)c->n_channels;
    HIP_TRY(hipMalloc((void **)&c->d_slip_buf, sizeof(uint32_t) * nch * (size_t)c->max_blocks * RDSP_BLOCK));
    for (auto &p : c->d_slip_carry) {
      HIP_TRY(hipMalloc((void **)&p, sizeof(uint32_t) * nch));
      HIP_TRY(hipMemset(p, 0, sizeof(uint32_t) * nch));
    }
  }
  c->iq_slip = slip;
  return RDSP_OK;
}
extern "C" int rdsp_sdr_setInputGain(rdsp_chain_t *c, float g) { NEED(c); TO_ENGINE(c, rdsp_engine_setInputGain(c->engine, g)); c->cfg.input_gain = g; return RDSP_OK; }
extern "C" int rdsp_sdr_setOutputGain(rdsp_chain_t *c, float g) { NEED(c); TO_ENGINE(c, rdsp_engine_setOutputGain(c->engine, g)); c->cfg.output_gain = g; return RDSP_OK; }
extern "C" int rdsp_sdr_setIQgainBalance(rdsp_chain_t *c, float g) { NEED(c); TO_ENGINE(c, rdsp_engine_setIQgainBalance(c->engine, g)); c->cfg.iq_balance = g; return RDSP_OK; }
extern "C" int rdsp_sdr_enableAudioFilter(rdsp_chain_t *c) {
  NEED(c);
  TO_ENGINE(c, rdsp_engine_enableAudioFilter(c->engine)); /* the engine's audio filter, not the CONV stage's bFilterEnabled */
  c->cfg.filter_on = 1;
  for (size_t i = 0; i < c->groups.size(); i++) {
    int rc = group_stage(c, (int)i);
    if (rc != RDSP_OK) return rc;
  }
  return RDSP_OK;
}
extern "C" int rdsp_sdr_setMute(rdsp_chain_t *c, int mute) { NEED(c); TO_ENGINE(c, rdsp_engine_setMute(c->engine, mute)); c->cfg.mute = mute ? 1 : 0; return RDSP_OK; }
extern "C" int rdsp_group_setTuningOffsetHz(rdsp_chain_t *c, int group, double hz) {
  if (check_group(c, group) != RDSP_OK) return RDSP_ERR_INVALID;
  c->groups[(size_t)group].nco_hz = hz;
  c->groups[(size_t)group].dirty = true;
  if (group == 0) c->cfg.nco_hz = hz;
  return RDSP_OK;
}
extern "C" int rdsp_sdr_setTuningOffsetHz(rdsp_chain_t *c, double hz) {
  NEED(c);
  if (c->engine) return RDSP_OK; /* the engine moves the carrier from its own offset to 0 Hz itself (INO:139, CTL:447) */
  for (size_t i = 0; i < c->groups.size(); i++) (void)rdsp_group_setTuningOffsetHz(c, (int)i, hz);
  return RDSP_OK;
}
extern "C" int rdsp_set_nr_level(rdsp_chain_t *c, int lvl) { NEED(c); c->cfg.lms_nr = lvl; return RDSP_OK; }
extern "C" int rdsp_set_spectral_nr(rdsp_chain_t *c, int on, float level) {
  NEED(c);
  if (on < 0 || on > 2) return RDSP_ERR_INVALID;
  c->cfg.spectral_nr = on;
  c->cfg.spectral_level = level;
  return RDSP_OK;
}

/* How both NLMS instances keep arm_lms_norm_f32's window energy.  0 (default): the reference's running difference
 * (`energy -= x0 * x0; energy += in * in`, NR:73) re-started from the exact 96-sample window sum at every 128-sample
 * block -- a deliberate deviation: after a loud-to-quiet transition the reference's own recursion can leave energy +
 * 1.19e-7 <= 0 and lose the channel.  1: the reference's arithmetic, one running difference for the whole stream,
 * for hosts that want NR:73 as it is, residue and all. */
extern "C" int rdsp_set_nlms_energy_mode(rdsp_chain_t *c, int running) {
  NEED(c);
  c->nlms_energy_running = running ? 1 : 0;
  return RDSP_OK;
}
/* SPEC:226-235 writes the re-synthesis as mag' (arm_cos_f32(phi) + j arm_sin_f32(phi)), phi = atan2(im, re).  0
 * (default): the exact-arithmetic equivalent X mag'/mag; 1: as written, with CMSIS' table-interpolated sine and
 * cosine as published (the two are 1.7e-5 - 1.9e-5 of the peak apart: the table's own interpolation error) */
extern "C" int rdsp_set_spectral_resynthesis(rdsp_chain_t *c, int literal) {
  NEED(c);
  if (literal && !c->d_sin_table) {
    if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
    float tab[513];
    rdsp_arm_sin_table(tab);
    HIP_TRY(hipMalloc((void **)&c->d_sin_table, sizeof(tab)));
    HIP_TRY(hipMemcpy(c->d_sin_table, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  c->spectral_literal = literal == 2 ? 2 : (literal ? 1 : 0);
  return RDSP_OK;
}

/* pass bands per audio filter and mode (CTL:149-191 names; Appendix C of the
 * survey: 150 Hz .. 2.1/2.7/3.1/3.9 kHz; CW 500 Hz wide around the 700 Hz pitch) */
static void passband(int filter, int demod, double *lo, double *hi) {
  double a = 150.0, b = 2700.0;
  switch (filter) {
    case RDSP_AUDIO_CW: a = 450.0; b = 950.0; break;
    case RDSP_AUDIO_2100: b = 2100.0; break;
    case RDSP_AUDIO_2700: b = 2700.0; break;
    case RDSP_AUDIO_3100: b = 3100.0; break;
    case RDSP_AUDIO_AM: b = 3900.0; break;
    case RDSP_AUDIO_WSPR: a = 1400.0; b = 1600.0; break;
    default: break;
  }
  if (demod == RDSP_DEMOD_LSB || demod == RDSP_DEMOD_CW_LSB) { *lo = -b; *hi = -a; }
  else if (demod == RDSP_DEMOD_AM || demod == RDSP_DEMOD_SAM) { *lo = -b; *hi = b; }
  else { *lo = a; *hi = b; }
}
/* the group's pass band under the current implementation of the audio filter: the mask carries
 * it (MASK), or the mask only selects the side band (50 Hz ... 4 kHz on the demodulator's side;
 * both sides for AM / SAM) and the band-pass is the group's biquad cascade (IIR) */
static int group_apply_audio_filter(rdsp_chain_t *c, int group, void *stream) {
  GroupState &g = c->groups[(size_t)group];
  double lo, hi;
  passband(g.audio_filter, g.demod, &lo, &hi);
  if (c->audio_kind == RDSP_AUDIO_KIND_IIR) {
    const double a = fabs(lo) < fabs(hi) ? fabs(lo) : fabs(hi), b = fabs(lo) < fabs(hi) ? fabs(hi) : fabs(lo);
    const double f1 = (g.demod == RDSP_DEMOD_AM || g.demod == RDSP_DEMOD_SAM) ? 150.0 : a;
    rdsp_design_audio_iir(f1, b, c->cfg.fs_in / (double)c->decim, g.iir);
    g.iir_dirty = true;
    if (g.demod == RDSP_DEMOD_LSB || g.demod == RDSP_DEMOD_CW_LSB) { lo = -4000.0; hi = -50.0; }
    else if (g.demod == RDSP_DEMOD_AM || g.demod == RDSP_DEMOD_SAM) { lo = -4000.0; hi = 4000.0; }
    else { lo = 50.0; hi = 4000.0; }
  }
  return rdsp_group_reInitializeFilter(c, group, lo, hi, stream);
}
extern "C" int rdsp_group_setAudioFilter(rdsp_chain_t *c, int group, int filter, void *stream) {
  if (check_group(c, group) != RDSP_OK) return RDSP_ERR_INVALID;
  if (filter < RDSP_AUDIO_CW || filter > RDSP_AUDIO_WSPR) return RDSP_ERR_INVALID;
  c->groups[(size_t)group].audio_filter = filter;
  return group_apply_audio_filter(c, group, stream);
}
/* which implementation SDR.setAudioFilter() selects filters of; re-applies every group's
 * current audio filter.  A control-path call: allocates the cascade's buffers on first use and
 * drains the tail stream (the cascade's state belongs to it). */
extern "C" int rdsp_sdr_setAudioFilterKind(rdsp_chain_t *c, int kind, void *stream) {
  NEED(c);
  if (kind != RDSP_AUDIO_KIND_MASK && kind != RDSP_AUDIO_KIND_IIR) return RDSP_ERR_INVALID;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (drain_tail_fwd(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (kind == RDSP_AUDIO_KIND_IIR) {
    const size_t nch = (size_t)c->n_channels, ng = c->groups.size();
    if (!c->d_iir_state) {
      HIP_TRY(hipMalloc((void **)&c->d_iir_state, sizeof(float) * 16 * nch));
      HIP_TRY(hipMemset(c->d_iir_state, 0, sizeof(float) * 16 * nch));
    }
    if (c->iir_sets < (int)ng) {
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      if (c->d_iir_coef) (void)hipFree(c->d_iir_coef);
      c->d_iir_coef = nullptr;
      HIP_TRY(hipMalloc((void **)&c->d_iir_coef, sizeof(float) * 20 * ng));
      c->iir_sets = (int)ng;
      for (auto &g : c->groups) g.iir_dirty = true;
    }
  }
  c->audio_kind = kind;
  for (size_t i = 0; i < c->groups.size(); i++) {
    int rc = group_apply_audio_filter(c, (int)i, stream);
    if (rc != RDSP_OK) return rc;
  }
  return RDSP_OK;
}
/* An explicit cascade for the group's audio filter instead of the designed one -- e.g. one of the engine's own
 * coefficient sets (the reference's firmware image holds fifteen of them): coef20 = four sections
 * {b0, b1, b2, a1, a2} in arm_biquad_cascade_df1_f32 order (feedback terms added).  Needs RDSP_AUDIO_KIND_IIR; the
 * mask keeps the side-band selection it has; the next setAudioFilter / setDemodMode designs a cascade again.
 * The sections' state is kept (a coefficient change mid-stream, like the sketch's filter menu). */
extern "C" int rdsp_group_setAudioIIRCoefficients(rdsp_chain_t *c, int group, const float *coef20) {
  if (check_group(c, group) != RDSP_OK || !coef20) return RDSP_ERR_INVALID;
  if (c->audio_kind != RDSP_AUDIO_KIND_IIR) {
    rdsp_set_error("rdsp_group_setAudioIIRCoefficients: select RDSP_AUDIO_KIND_IIR first (rdsp_sdr_setAudioFilterKind)");
    return RDSP_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < 20; i++)
    if (!(coef20[i] == coef20[i]) || fabsf(coef20[i]) > 1e6f) {
      rdsp_set_error("rdsp_group_setAudioIIRCoefficients: coefficient %d is not a finite filter coefficient", i);
      return RDSP_ERR_INVALID;
    }
  GroupState &g = c->groups[(size_t)group];
  memcpy(g.iir, coef20, sizeof(g.iir));
  g.iir_dirty = true;
  return RDSP_OK;
}
extern "C" int rdsp_sdr_setAudioIIRCoefficients(rdsp_chain_t *c, const float *coef20) {
  NEED(c);
  for (size_t i = 0; i < c->groups.size(); i++) {
    int rc = rdsp_group_setAudioIIRCoefficients(c, (int)i, coef20);
    if (rc != RDSP_OK) return rc;
  }
  return RDSP_OK;
}
extern "C" int rdsp_chain_get_iir_coeffs(rdsp_chain_t *c, int group, float *out20) {
  if (check_group(c, group) != RDSP_OK || !out20) return RDSP_ERR_INVALID;
  memcpy(out20, c->groups[(size_t)group].iir, sizeof(float) * 20);
  return RDSP_OK;
}
extern "C" int rdsp_sdr_setAudioFilter(rdsp_chain_t *c, int filter, void *stream) {
  NEED(c);
  if (c->engine) {
    const int id = engine_filter_of(filter);
    if (id < 0) { rdsp_set_error("rdsp_sdr_setAudioFilter: no engine filter id known for %d", filter); return RDSP_ERR_INVALID; }
    return rdsp_engine_setAudioFilter(c->engine, id);
  }
  for (size_t i = 0; i < c->groups.size(); i++) {
    int rc = rdsp_group_setAudioFilter(c, (int)i, filter, stream);
    if (rc != RDSP_OK) return rc;
  }
  return RDSP_OK;
}
extern "C" uint32_t rdsp_group_setDemodMode(rdsp_chain_t *c, int group, int mode, void *stream) {
  if (check_group(c, group) != RDSP_OK || mode < RDSP_DEMOD_IQ || mode > RDSP_DEMOD_SAM) return 0;
  GroupState &g = c->groups[(size_t)group];
  if (mode == RDSP_DEMOD_SAM && ensure_sam(c) != RDSP_OK) return 0;
  g.demod = mode;
  if (group == 0) c->cfg.demod = mode;
  (void)group_apply_audio_filter(c, group, stream);
  return demod_tuning_offset(mode);
}
extern "C" uint32_t rdsp_sdr_setDemodMode(rdsp_chain_t *c, int mode, void *stream) {
  if (!c || mode < RDSP_DEMOD_IQ || mode > RDSP_DEMOD_SAM) return 0;
  if (c->engine) return engine_mode_of(mode) < 0 ? 0u : (uint32_t)rdsp_engine_setDemodMode(c->engine, engine_mode_of(mode));
  for (size_t i = 0; i < c->groups.size(); i++) (void)rdsp_group_setDemodMode(c, (int)i, mode, stream);
  return demod_tuning_offset(mode);
}

/* ---- receiver groups ---------------------------------------------------- */
extern "C" int rdsp_chain_groups(const rdsp_chain_t *c) { return c ? (int)c->groups.size() : 0; }

/* partition the channels into n_groups receiver groups; group_of_channel[ch] < n_groups
 * (NULL with n_groups == 1 restores the single shared group).  Synchronises the
 * device: a set-up call, not a streaming one.  New groups start as copies of group 0. */
extern "C" int rdsp_chain_set_groups(rdsp_chain_t *c, int n_groups, const uint16_t *group_of_channel) {
  NEED(c);
  if (n_groups < 1 || n_groups > 65535 || (n_groups > 1 && !group_of_channel)) {
    rdsp_set_error("rdsp_chain_set_groups: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (group_of_channel)
    for (int i = 0; i < c->n_channels; i++)
      if (group_of_channel[i] >= n_groups) {
        rdsp_set_error("channel %d: group %d >= n_groups %d", i, (int)group_of_channel[i], n_groups);
        return RDSP_ERR_INVALID;
      }
  if (drain_tail_fwd(c) != RDSP_OK) return RDSP_ERR_HIP;
  /* documented as synchronising: the caller's processing stream is not known here and may be a
   * non-blocking one (no null-stream sync covers it), and the group tables below are freed and
   * reallocated -- every kernel that may still read them has finished after this */
  HIP_TRY(hipDeviceSynchronize());
  int rc = groups_resize(c, n_groups);
  if (rc != RDSP_OK) return rc;
  if (c->d_group_of) { (void)hipFree(c->d_group_of); c->d_group_of = nullptr; }
  c->group_of.clear();
  if (group_of_channel && n_groups > 1) {
    c->group_of.assign(group_of_channel, group_of_channel + c->n_channels);
    HIP_TRY(hipMalloc((void **)&c->d_group_of, sizeof(uint16_t) * (size_t)c->n_channels));
    HIP_TRY(hipMemcpy(c->d_group_of, c->group_of.data(), sizeof(uint16_t) * (size_t)c->n_channels, hipMemcpyHostToDevice));
  }
  /* with the IIR bank selected, re-selecting it stages every group's mask (side-band selector) and
   * coefficient set; otherwise the masks alone */
  if (c->audio_kind == RDSP_AUDIO_KIND_IIR) return rdsp_sdr_setAudioFilterKind(c, RDSP_AUDIO_KIND_IIR, nullptr);
  for (int g = 0; g < n_groups; g++) {
    rc = group_stage(c, g);
    if (rc != RDSP_OK) return rc;
  }
  return RDSP_OK;
}

extern "C" int rdsp_group_get_mask(rdsp_chain_t *c, int group, float *host_out) {
  if (check_group(c, group) != RDSP_OK || !host_out) return RDSP_ERR_INVALID;
  memcpy(host_out, c->groups[(size_t)group].mask_nat.data(), sizeof(float) * 2 * (size_t)c->N);
  return RDSP_OK;
}

/* checkPBT_Increase / checkPBT_Decrease (CTL:569-612) on a pair of cut-offs:
 * edge 0 = LOCUT (button D3), 1 = HICUT (D6); dir +1 / -1; 50 Hz steps inside
 * [MIN_LOW, MAX_LOW] and [MIN_HI, MAX_HI] (RDSP_general_includes.h:79-82), with the
 * reference's comparisons (<= when increasing, > when decreasing). */
extern "C" int rdsp_pbt_step(double *lo, double *hi, int edge, int dir) {
  if (!lo || !hi || (edge != 0 && edge != 1) || (dir != 1 && dir != -1)) return RDSP_ERR_INVALID;
  const double MIN_LOW = 0.0, MAX_LOW = 700.0, MIN_HI = 800.0, MAX_HI = 4000.0;
  if (dir > 0) {
    if (edge == 0) *lo = (*lo + 50) <= MAX_LOW ? (*lo + 50) : *lo; /* CTL:574 */
    else *hi = (*hi + 50) <= MAX_HI ? (*hi + 50) : *hi;            /* CTL:581 */
  } else {
    if (edge == 0) {
      *lo = (*lo - 50) > MIN_LOW ? (*lo - 50) : *lo; /* CTL:595 */
      if (*lo < 0.0) *lo = 0.0;                      /* CTL:596 */
    } else {
      *hi = (*hi - 50) > MIN_HI ? (*hi - 50) : *hi;  /* CTL:604 */
    }
  }
  return RDSP_OK;
}
extern "C" int rdsp_group_pbt(rdsp_chain_t *c, int group, int edge, int dir, void *stream) {
  if (check_group(c, group) != RDSP_OK) return RDSP_ERR_INVALID;
  GroupState &g = c->groups[(size_t)group];
  double lo = g.lo, hi = g.hi;
  int rc = rdsp_pbt_step(&lo, &hi, edge, dir);
  if (rc != RDSP_OK) return rc;
  return rdsp_group_reInitializeFilter(c, group, lo, hi, stream); /* CTL:575,582,597,605 */
}

/* tuningMode() (CTL:330-423): the mode table of the sketch.  mndx 0 "CW N" (500 Hz),
 * 1 "CW" (2.1 kHz), 2 "USB", 3 "LSB", 4 "AM", 5 "SAM", 6 "RTTY"; CW side chosen by
 * vfoFreq > 10 MHz (CTL:337,349).  Returns TuningOffset. */
extern "C" uint32_t rdsp_group_tuningMode(rdsp_chain_t *c, int group, int mndx, double vfo_hz, void *stream) {
  if (check_group(c, group) != RDSP_OK) return 0;
  int filter, mode;
  switch (mndx) {
    case 0: filter = RDSP_AUDIO_CW; mode = vfo_hz > 10000000.0 ? RDSP_DEMOD_CW_USB : RDSP_DEMOD_CW_LSB; break;
    case 1: filter = RDSP_AUDIO_2100; mode = vfo_hz > 10000000.0 ? RDSP_DEMOD_CW_USB : RDSP_DEMOD_CW_LSB; break;
    case 2: filter = RDSP_AUDIO_2700; mode = RDSP_DEMOD_USB; break;
    case 3: filter = RDSP_AUDIO_2700; mode = RDSP_DEMOD_LSB; break;
    case 4: filter = RDSP_AUDIO_AM; mode = RDSP_DEMOD_AM; break;
    case 5: filter = RDSP_AUDIO_AM; mode = RDSP_DEMOD_SAM; break;
    case 6: filter = RDSP_AUDIO_2100; mode = RDSP_DEMOD_USB; break;
    default: rdsp_set_error("tuningMode: no menu entry %d (CTL:330-423 has 0..6)", mndx); return 0;
  }
  c->groups[(size_t)group].audio_filter = filter;      /* SDR.setAudioFilter(...) */
  return rdsp_group_setDemodMode(c, group, mode, stream); /* TuningOffset = SDR.setDemodMode(...) */
}

/* ---- pipelined mode ---------------------------------------------------------------- */
/* everything queued on the internal tail stream has finished when this returns */
static int drain_tail(rdsp_chain_t *c) {
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  return RDSP_OK;
}
static int drain_tail_fwd(rdsp_chain_t *c) { return drain_tail(c); }
extern "C" int rdsp_chain_set_pipelined(rdsp_chain_t *c, int on) {
  NEED(c);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (drain_tail(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (on && !c->s_tail) {
    /* (a queue priority on this stream, hipStreamCreateWithPriority high or low, changes nothing: K3 1.168-1.192 /
     * 1.165-1.178 against 1.163-1.190 ms, K5 2.305-2.316 / 2.323-2.415 against 2.293-2.335; round 5, same box) */
    HIP_TRY(hipStreamCreateWithFlags(&c->s_tail, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->s_mid, hipStreamNonBlocking));
    for (int i = 0; i < 3; i++) HIP_TRY(hipEventCreateWithFlags(&c->ev_mid[i], hipEventDisableTiming));
    for (int i = 0; i < 3; i++) {
      HIP_TRY(hipEventCreateWithFlags(&c->ev_front[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&c->ev_tail[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_misc, hipEventDisableTiming));
    for (int i = 0; i < 2; i++)
      HIP_TRY(hipMalloc((void **)&c->d_midx[i], sizeof(float) * c->mid_stride * (size_t)c->n_channels));
  }
  c->pipe_on = on ? 1 : 0;
  c->call_idx = 0;
  c->tail_slot = -1; /* drained above */
  return on ? ensure_sub_batch_events(c) : RDSP_OK;
}
/* one event per channel sub-batch and intermediate buffer; made here and in
 * rdsp_chain_set_sub_batch, never on the streaming path */
static int ensure_sub_batch_events(rdsp_chain_t *c) {
  if (!c->s_tail || c->sub_batch <= 0) return RDSP_OK;
  const size_t nsb = (size_t)((c->n_channels + c->sub_batch - 1) / c->sub_batch);
  for (int slot = 0; slot < 3; slot++)
    while (c->ev_front_sb[slot].size() < nsb) {
      hipEvent_t ev;
      HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      c->ev_front_sb[slot].push_back(ev);
    }
  return RDSP_OK;
}
/* PLL state and the quadrature intermediates of the SAM demodulator: allocated when a group is
 * first switched to SAMmode (a control-path call), not by the processing call */
static int ensure_sam(rdsp_chain_t *c) {
  if (c->d_sam) return RDSP_OK;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  const size_t nch = (size_t)c->n_channels, mid_bytes = sizeof(float) * c->mid_stride * nch;
  for (int i = 0; i < 3; i++) HIP_TRY(hipMalloc((void **)&c->d_mid_q[i], mid_bytes));
  HIP_TRY(hipMalloc((void **)&c->d_sam, sizeof(float) * 4 * nch));
  HIP_TRY(hipMemset(c->d_sam, 0, sizeof(float) * 4 * nch));
  return RDSP_OK;
}
/* front-kernel variant: -1 = auto (full-register; measured faster with and without the
 * concurrent tail stage), 0 = full-register, 1 = lean (FFT twiddles rebuilt per pass).  Both compute the same chain; they differ in the
 * rounding of the FFT twiddles (power chain vs direct), ~3e-7. */
extern "C" int rdsp_chain_set_front_variant(rdsp_chain_t *c, int lean) {
  NEED(c);
  if (lean < -1 || lean > 1) return RDSP_ERR_INVALID;
  c->lean_mode = lean;
  return RDSP_OK;
}
/* pipelined calls are launched in channel sub-batches of this size (a multiple of 64; 0 = one
 * launch per stage whatever the channel count).  Results do not depend on it. */
extern "C" int rdsp_chain_set_sub_batch(rdsp_chain_t *c, int channels) {
  NEED(c);
  if (channels < 0 || channels % 64 != 0) return RDSP_ERR_INVALID;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  c->sub_batch = channels;
  return ensure_sub_batch_events(c);
}
/* wave priorities (s_setprio 0..3) used while the tail stage shares the SIMDs with the front
 * stage of the next call: the front kernel's during its FIR, the tail kernel's throughout */
extern "C" int rdsp_chain_set_priorities(rdsp_chain_t *c, int front_fir_prio, int tail_prio) {
  NEED(c);
  if (front_fir_prio < 0 || front_fir_prio > 3 || tail_prio < 0 || tail_prio > 3) return RDSP_ERR_INVALID;
  c->front_fir_prio = front_fir_prio;
  c->tail_prio = tail_prio;
  return RDSP_OK;
}
/* stage A3 of the front kernel (decim 4; decim-1 chains have no decimator and always run rdsp_front_kernel).
 * -1 (default): 4, or 5 for calls that no tail / SAM / IIR stage follows and that run without the blanker (rdsp_chain_process).
 * 4: in the frequency domain -- polyphase overlap-save: four low-rate transforms, branch
 * spectra, one inverse -- with frames of one granule (256 outputs; the rest of the 512-point window zeros): every
 * call boundary is a frame boundary and every frame's input is a function of the absolute sample position, so a
 * stream gives the same bits however it is cut into calls, like the reference's fixed 128-sample blocks
 * (CONV:231-245).  0: the direct form (packed FMAs), split-invariant too, ~1.3x slower.  2: the frequency domain
 * with 448-sample frames anchored at each call's first sample: 5 transforms per 448 outputs instead of per 256,
 * but a different call split frames and rounds differently (~3e-7): the throughput form, what bench.py selects.
 * 5: the frequency domain on 16-lane rows -- 256-point windows, four per wave, 128 outputs each (two frames per
 * granule): split-invariant like the default, ~10 % faster than it for chains without a tail stage (K2 0.727 against
 * 0.808 ms), no gain beside a tail kernel (250 registers); with the noise blanker on it runs the default form.
 * Same taps and the same exact linear convolution in all of them; the sums associate differently (~2e-7).
 * EXPERIMENTAL=1 builds: 1 = v_mfma GEMM slices, 3 = the same unless the tail stage runs concurrently, 6 = the row
 * form with 192 outputs per window (frames anchored at the call's first sample; measured, no gain over 2). */
extern "C" int rdsp_chain_set_fir_variant(rdsp_chain_t *c, int variant) {
  NEED(c);
  if (variant < -1 || variant > 6) return RDSP_ERR_INVALID;
  if ((variant == 2 || variant >= 4) && !c->d_fd_mask) {
    rdsp_set_error("the frequency-domain decimator needs decim = 4");
    return RDSP_ERR_UNSUPPORTED;
  }
#ifndef RDSP_EXPERIMENTAL
  if (variant == 1 || variant == 3) {
    rdsp_set_error("the matrix-core FIR is only in EXPERIMENTAL=1 builds of the library");
    return RDSP_ERR_UNSUPPORTED;
  }
  if (variant == 6) {
    rdsp_set_error("the row form with 192 outputs per window is only in EXPERIMENTAL=1 builds of the library");
    return RDSP_ERR_UNSUPPORTED;
  }
#endif
  c->fir_mode = variant;
  return RDSP_OK;
}
/* tail-kernel variant.  (16, 2) is the product (rdsp_tail.hip: a channel per 16-lane DPP row, two
 * steps per reduction).  EXPERIMENTAL=1 builds: (16, 4) weights one block stale with a hand-interleaved
 * issue order, (16, 5) four steps per reduction (both round 3), (16, 3) one reduction per step (round 1), (8, 2) half a row per channel,
 * (16 | 8, 1) the reduction on the matrix pipe, (16, 0) the delay line shifted by DPP.  All compute
 * the same recursion; the sums associate differently. */
extern "C" int rdsp_chain_set_tail_variant(rdsp_chain_t *c, int lanes_per_channel, int matrix_reduce) {
  NEED(c);
  if ((lanes_per_channel != 8 && lanes_per_channel != 16) || (lanes_per_channel == 8 && !matrix_reduce) ||
      matrix_reduce < 0 || matrix_reduce > 5 || (lanes_per_channel == 8 && matrix_reduce > 2))
    return RDSP_ERR_INVALID;
  int v;
  if (lanes_per_channel == 16 && matrix_reduce == 2) v = 100;
  else {
#ifndef RDSP_EXPERIMENTAL
    rdsp_set_error("tail-kernel variants other than the product's are only in EXPERIMENTAL=1 builds of the library");
    return RDSP_ERR_UNSUPPORTED;
#else
    if (matrix_reduce == 5) v = 105;
    else if (matrix_reduce == 4) v = 104;
    else if (matrix_reduce == 3) v = 102;
    else if (matrix_reduce == 2) v = 101;
    else v = lanes_per_channel + (matrix_reduce ? 100 : 0);
#endif
  }
  if (drain_tail(c) != RDSP_OK) return RDSP_ERR_HIP;
  c->tail_lpc = v;
  return RDSP_OK;
}
/* `stream` waits for every call issued so far (outputs complete after it) */
extern "C" int rdsp_chain_flush(rdsp_chain_t *c, void *stream) {
  NEED(c);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (c->tail_slot >= 0) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, c->ev_tail[c->tail_slot], 0));
  return RDSP_OK;
}

/* name of the front kernel the most recent rdsp_chain_process launched (as the profiler shows it,
 * without template arguments): rdsp_front_fd_kernel or rdsp_front_kernel */
extern "C" const char *rdsp_chain_front_kernel_name(const rdsp_chain_t *c) { return c ? c->front_name : ""; }

/* ---- per-kernel timing with HIP events on the launch stream -------------------- */
extern "C" int rdsp_chain_set_timing(rdsp_chain_t *c, int on) {
  NEED(c);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  const size_t kMaxCalls = 1024; /* calls beyond the pool are simply not timed */
  if (on && c->ev.empty()) {
    c->ev.resize(4 * kMaxCalls);
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    c->ev_has_tail.assign(kMaxCalls, 0);
  }
  c->ev_used = 0;
  c->timing_on = on ? 1 : 0;
  return RDSP_OK;
}
/* sums over the calls recorded since rdsp_chain_set_timing(c, 1) */
extern "C" int rdsp_chain_get_timing(rdsp_chain_t *c, double *front_ms, double *tail_ms, int *calls) {
  NEED(c);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  double f = 0.0, t = 0.0;
  const size_t n = c->ev_used;
  for (size_t i = 0; i < n; i++) {
    float a = 0.f, b = 0.f;
    HIP_TRY(hipEventSynchronize(c->ev[4 * i + 1]));
    HIP_TRY(hipEventElapsedTime(&a, c->ev[4 * i], c->ev[4 * i + 1]));
    f += a;
    if (c->ev_has_tail[i]) {
      HIP_TRY(hipEventSynchronize(c->ev[4 * i + 3]));
      HIP_TRY(hipEventElapsedTime(&b, c->ev[4 * i + 2], c->ev[4 * i + 3]));
      t += b;
    }
  }
  if (front_ms) *front_ms = f;
  if (tail_ms) *tail_ms = t;
  if (calls) *calls = (int)n;
  return RDSP_OK;
}

/* milliseconds from the end of the first recorded call's last kernel to the end of the last recorded call's:
 * (calls - 1) steady-state periods of a pipelined sequence, without the pipeline's fill (the first call's
 * front kernel has no tail kernel to overlap with) */
extern "C" int rdsp_chain_get_timing_span(rdsp_chain_t *c, double *span_ms, int *calls) {
  NEED(c);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  const size_t n = c->ev_used;
  float ms = 0.f;
  if (n >= 2) {
    hipEvent_t a = c->ev[4 * 0 + (c->ev_has_tail[0] ? 3 : 1)];
    hipEvent_t b = c->ev[4 * (n - 1) + (c->ev_has_tail[n - 1] ? 3 : 1)];
    HIP_TRY(hipEventSynchronize(b));
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
  }
  if (span_ms) *span_ms = ms;
  if (calls) *calls = (int)n;
  return RDSP_OK;
}

/* ---- per-channel state as data: checkpoint / resume, channels moved between chains or GPUs ------
 * The reference keeps its DSP state in globals (CONV:50-57,77-80; NR:26-32; SPEC:109) and has no
 * persistence; here the state is an explicit per-channel record, so a range of
 * channels can be written out and read back into the same range of another chain -- same FFT_L and
 * decimation, same settings (modes, filters and gains are configuration: the caller re-applies them).
 * Blob: header, then the arrays of DESIGN.md 3 for the n channels, each [n][...]. */
namespace {
struct StateHeader {
  uint32_t magic, version; /* "RDSP", 4 */
  int32_t n_channels, fft_l, decim;
  int32_t has_sam, has_iir;
  int32_t old_nr_level;
  uint64_t n_in;
  int64_t nr_calls, als_calls;
  float nr_mu, als_mu;
  int32_t hist_valid, hist_swap;
  float hist_scale_i, hist_scale_q;
  int32_t n_groups;      /* followed by n_groups x {has_dev_dphi, dev_dphi}: the NCO increment each group's FIR
                            history was mixed with (a tuning change right before the checkpoint) */
  int32_t has_slip;      /* the last call ran with the I2S slip correction: its carry word travels too */
  int32_t fir_fd;        /* stage A3 of the saving chain: 0 direct, 1 frequency domain with 448-sample frames, 2 with
                            granule frames (informative: all keep the same 256 raw samples, so a stream may be
                            continued in any of them) */
};
constexpr uint32_t kStateVersion = 4;
constexpr uint32_t kStateMagic = 0x50534452u; /* 'R' 'D' 'S' 'P' */
struct StatePart { void *dev; size_t per_channel; };
/* the per-channel arrays in blob order; optional ones (SAM, IIR, slip carry) only when present */
std::vector<StatePart> state_parts(const rdsp_chain_t *c, bool sam, bool iir, bool slip = false) {
  std::vector<StatePart> v = {
      {c->d_hist, sizeof(uint32_t) * 256},
      {c->d_prev, sizeof(float2) * (size_t)(c->N / 2)},
      {c->d_scal, sizeof(float) * 4},
      {c->d_nr_w, sizeof(float) * RDSP_LMS_TAPS}, {c->d_nr_prev, sizeof(float) * RDSP_BLOCK}, {c->d_nr_energy, sizeof(float)},
      {c->d_als_w, sizeof(float) * RDSP_LMS_TAPS}, {c->d_als_prev, sizeof(float) * RDSP_BLOCK}, {c->d_als_energy, sizeof(float)},
      {c->d_status, sizeof(uint32_t)}, {c->d_status + c->n_channels, sizeof(uint32_t)}, /* health words: DSP-NR, ALS */
  };
  if (sam) v.push_back({c->d_sam, sizeof(float) * 4});
  if (iir) v.push_back({c->d_iir_state, sizeof(float) * 16});
  if (slip) v.push_back({c->d_slip_carry[c->slip_phase], sizeof(uint32_t)});
  return v;
}
size_t state_bytes(const rdsp_chain_t *c, int n, bool sam, bool iir, bool slip, size_t n_groups) {
  size_t b = sizeof(StateHeader) + 2 * sizeof(uint32_t) * n_groups;
  for (const auto &p : state_parts(c, sam, iir, slip)) b += p.per_channel * (size_t)n;
  return b;
}
}  // namespace

extern "C" size_t rdsp_chain_state_bytes(const rdsp_chain_t *c, int n_channels) {
  if (!c || n_channels <= 0 || n_channels > c->n_channels) return 0;
  /* an upper bound that only set-up calls change: optional parts count once their buffers exist (the slip
   * carry travels only when the last call ran corrected, but its place is reserved as soon as
   * rdsp_pre_setIQslip has allocated it), so a buffer sized after set-up fits every later save */
  return state_bytes(c, n_channels, c->d_sam != nullptr, c->d_iir_state != nullptr, c->d_slip_buf != nullptr, c->groups.size());
}

/* everything queued so far has finished when the copy is taken (a control-path call) */
extern "C" int rdsp_chain_save_state(rdsp_chain_t *c, int first_channel, int n_channels, void *host_buf, size_t bytes,
                                     void *stream) {
  NEED(c);
  if (!host_buf || first_channel < 0 || n_channels <= 0 || first_channel + n_channels > c->n_channels ||
      bytes < rdsp_chain_state_bytes(c, n_channels)) {
    rdsp_set_error("rdsp_chain_save_state: bad argument (channels %d..%d of %d, %zu bytes, %zu needed)", first_channel,
                   first_channel + n_channels, c->n_channels, bytes, rdsp_chain_state_bytes(c, n_channels));
    return RDSP_ERR_INVALID;
  }
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  StateHeader h;
  memset(&h, 0, sizeof(h));
  h.magic = kStateMagic; h.version = kStateVersion;
  h.has_slip = c->slip_prev_on ? 1 : 0;
  h.fir_fd = fir_fd_of(c); /* decim 1: no decimator */
  h.n_channels = n_channels; h.fft_l = c->N; h.decim = c->decim;
  h.has_sam = c->d_sam != nullptr; h.has_iir = c->d_iir_state != nullptr;
  h.old_nr_level = c->old_nr_level; h.n_in = c->n_in;
  h.nr_calls = c->nr_calls; h.als_calls = c->als_calls;
  h.nr_mu = c->nr_mu; h.als_mu = c->als_mu;
  h.hist_valid = c->hist_valid; h.hist_swap = c->hist_swap;
  h.hist_scale_i = c->hist_scale_i; h.hist_scale_q = c->hist_scale_q;
  h.n_groups = (int32_t)c->groups.size();
  unsigned char *dst = (unsigned char *)host_buf;
  memcpy(dst, &h, sizeof(h));
  dst += sizeof(h);
  for (const auto &g : c->groups) {
    const uint32_t w[2] = {g.has_dev_dphi ? 1u : 0u, g.dev_dphi};
    memcpy(dst, w, sizeof(w));
    dst += sizeof(w);
  }
  for (const auto &p : state_parts(c, h.has_sam, h.has_iir, h.has_slip)) {
    const size_t n = p.per_channel * (size_t)n_channels;
    HIP_TRY(hipMemcpy(dst, (const unsigned char *)p.dev + p.per_channel * (size_t)first_channel, n, hipMemcpyDeviceToHost));
    dst += n;
  }
  return RDSP_OK;
}

/* the blob's channels become channels first_channel .. of this chain.  A chain that has not processed
 * anything yet also takes the stream position and the call history (resume); one that has must be at
 * the same stream position (channels moved between shards of one stream). */
extern "C" int rdsp_chain_load_state(rdsp_chain_t *c, int first_channel, const void *host_buf, size_t bytes, void *stream) {
  NEED(c);
  StateHeader h;
  if (!host_buf || bytes < sizeof(h)) {
    rdsp_set_error("rdsp_chain_load_state: bad argument");
    return RDSP_ERR_INVALID;
  }
  memcpy(&h, host_buf, sizeof(h));
  if (h.magic != kStateMagic || h.version != kStateVersion || h.fft_l != c->N || h.decim != c->decim || h.n_channels <= 0 ||
      h.n_groups < 1 || first_channel < 0 || first_channel + h.n_channels > c->n_channels) {
    rdsp_set_error("rdsp_chain_load_state: blob (version %u) of %d channels, FFT_L %d, decimation %d does not fit channels %d.. "
                   "of a chain of %d channels, FFT_L %d, decimation %d (blob version %u)", h.version, h.n_channels, h.fft_l,
                   h.decim, first_channel, c->n_channels, c->N, c->decim, kStateVersion);
    return RDSP_ERR_INVALID;
  }
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  /* A stream continued from a blob is the uninterrupted stream bit for bit, or the call fails: nothing
   * is restored in part.  Optional state the blob carries must have a place in this chain. */
  if (h.has_iir && !c->d_iir_state) {
    rdsp_set_error("rdsp_chain_load_state: the blob carries the IIR audio filter's state; select it first "
                   "(rdsp_sdr_setAudioFilterKind(chain, RDSP_AUDIO_KIND_IIR))");
    return RDSP_ERR_INVALID;
  }
  if (h.has_slip && !c->d_slip_buf) {
    rdsp_set_error("rdsp_chain_load_state: the blob was taken with the I2S slip correction on; call rdsp_pre_setIQslip first");
    return RDSP_ERR_INVALID;
  }
  if (h.has_sam && ensure_sam(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (bytes < state_bytes(c, h.n_channels, h.has_sam != 0, h.has_iir != 0, h.has_slip != 0, (size_t)h.n_groups)) {
    rdsp_set_error("rdsp_chain_load_state: blob truncated");
    return RDSP_ERR_INVALID;
  }
  const unsigned char *gsrc = (const unsigned char *)host_buf + sizeof(h);
  const bool fresh = c->n_in == 0 && c->call_idx == 0;
  if ((size_t)h.n_groups != c->groups.size()) { /* another partition: fine unless a history increment would be lost */
    for (int g = 0; g < h.n_groups; g++) {
      uint32_t w[2];
      memcpy(w, gsrc + 2 * sizeof(uint32_t) * (size_t)g, sizeof(w));
      if (w[0]) {
        rdsp_set_error("rdsp_chain_load_state: the blob has %d receiver groups with a tuning change pending in a FIR history, "
                       "the chain %zu groups: set the same groups first", h.n_groups, c->groups.size());
        return RDSP_ERR_INVALID;
      }
    }
  }
  if (!fresh) { /* channels moved between shards of one stream: both sides must be at the same point of it */
    const bool same = c->n_in == h.n_in && (c->nr_calls == 0) == (h.nr_calls == 0) && (c->als_calls == 0) == (h.als_calls == 0) &&
                      c->hist_valid == (h.hist_valid != 0) && c->hist_swap == h.hist_swap && c->hist_scale_i == h.hist_scale_i &&
                      c->hist_scale_q == h.hist_scale_q && c->old_nr_level == h.old_nr_level && c->slip_prev_on == (h.has_slip != 0);
    if (!same) {
      rdsp_set_error("rdsp_chain_load_state: the chain (input sample %llu) and the blob (input sample %llu) are not at the same "
                     "point of the stream / call history", (unsigned long long)c->n_in, (unsigned long long)h.n_in);
      return RDSP_ERR_INVALID;
    }
  }
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  const unsigned char *src = gsrc + 2 * sizeof(uint32_t) * (size_t)h.n_groups;
  for (const auto &p : state_parts(c, h.has_sam != 0, h.has_iir != 0, h.has_slip != 0)) {
    const size_t n = p.per_channel * (size_t)h.n_channels;
    HIP_TRY(hipMemcpy((unsigned char *)p.dev + p.per_channel * (size_t)first_channel, src, n, hipMemcpyHostToDevice));
    src += n;
  }
  /* optional state the chain has and the blob does not starts from zero for these channels, like a fresh chain's */
  if (!h.has_sam && c->d_sam) HIP_TRY(hipMemset(c->d_sam + 4 * (size_t)first_channel, 0, sizeof(float) * 4 * (size_t)h.n_channels));
  if (!h.has_iir && c->d_iir_state)
    HIP_TRY(hipMemset(c->d_iir_state + 16 * (size_t)first_channel, 0, sizeof(float) * 16 * (size_t)h.n_channels));
  if (fresh) {
    c->n_in = h.n_in;
    c->old_nr_level = h.old_nr_level;
    c->nr_calls = (long)h.nr_calls; c->als_calls = (long)h.als_calls;
    c->nr_mu = h.nr_mu; c->als_mu = h.als_mu;
    c->hist_valid = h.hist_valid != 0; c->hist_swap = h.hist_swap;
    c->hist_scale_i = h.hist_scale_i; c->hist_scale_q = h.hist_scale_q;
    c->slip_prev_on = h.has_slip != 0;
    if ((size_t)h.n_groups == c->groups.size()) /* same partition: the increments the histories came in with */
      for (auto &g : c->groups) {
        uint32_t w[2];
        memcpy(w, gsrc, sizeof(w));
        gsrc += sizeof(w);
        g.has_dev_dphi = w[0] != 0;
        g.dev_dphi = w[1];
        g.dirty = true;
      }
  }
  return RDSP_OK;
}

/* ---- state read-back ------------------------------------------------------- */
extern "C" int rdsp_chain_get_scalars(rdsp_chain_t *c, float *host_out, void *stream) {
  NEED(c);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  HIP_TRY(hipMemcpy(host_out, c->d_scal, sizeof(float) * 4 * (size_t)c->n_channels, hipMemcpyDeviceToHost));
  return RDSP_OK;
}
/* A channel whose NLMS instance has run away (rdsp_chain_get_status) stays dead: arm_lms_norm_init_f32
 * leaves the coefficients (NR:62), so Init_LMS_NR does not clear infinite weights, and the sketch's only
 * cure is a power cycle.  With thousands of receivers the host clears just the ones that need it: the
 * instance's weights, delay block, energy and health word of channels [first, first + count) go back to
 * their boot values, in stream order behind everything queued so far; no other channel is touched. */
extern "C" int rdsp_chain_reset_nlms_channels(rdsp_chain_t *c, int which, int first_channel, int n_channels, void *stream_) {
  NEED(c);
  if ((which != 0 && which != 1) || first_channel < 0 || n_channels <= 0 || first_channel + n_channels > c->n_channels) {
    rdsp_set_error("rdsp_chain_reset_nlms_channels: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  hipStream_t stream = (hipStream_t)stream_;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail)); /* the tail stage owns these arrays */
  const size_t f = (size_t)first_channel, n = (size_t)n_channels, nch = (size_t)c->n_channels;
  float *w = which ? c->d_als_w : c->d_nr_w, *prev = which ? c->d_als_prev : c->d_nr_prev, *en = which ? c->d_als_energy : c->d_nr_energy;
  HIP_TRY(hipMemsetAsync(w + RDSP_LMS_TAPS * f, 0, sizeof(float) * RDSP_LMS_TAPS * n, stream));
  HIP_TRY(hipMemsetAsync(prev + RDSP_BLOCK * f, 0, sizeof(float) * RDSP_BLOCK * n, stream));
  HIP_TRY(hipMemsetAsync(en + f, 0, sizeof(float) * n, stream));
  HIP_TRY(hipMemsetAsync(c->d_status + (which ? nch : 0) + f, 0, sizeof(uint32_t) * n, stream));
  if (c->s_tail) {
    HIP_TRY(hipEventRecord(c->ev_misc, stream));
    HIP_TRY(hipStreamWaitEvent(c->s_tail, c->ev_misc, 0));
  }
  return RDSP_OK;
}
/* per-channel health word: RDSP_STATUS_* bits, sticky until rdsp_Init_LMS_NR (DSP-NR bits) / rdsp_chain_reset */
extern "C" int rdsp_chain_get_status(rdsp_chain_t *c, uint32_t *host_out, void *stream) {
  NEED(c);
  if (!host_out) return RDSP_ERR_INVALID;
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  const size_t nch = (size_t)c->n_channels;
  std::vector<uint32_t> w(2 * nch);
  HIP_TRY(hipMemcpy(w.data(), c->d_status, sizeof(uint32_t) * 2 * nch, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nch; i++) host_out[i] = (w[i] & 3u) | ((w[nch + i] & 3u) << 4);
  return RDSP_OK;
}
extern "C" int rdsp_chain_get_lms_coeffs(rdsp_chain_t *c, int which, float *host_out, void *stream) {
  NEED(c);
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  if (c->s_tail) HIP_TRY(hipStreamSynchronize(c->s_tail));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  HIP_TRY(hipMemcpy(host_out, which ? c->d_als_w : c->d_nr_w,
                    sizeof(float) * RDSP_LMS_TAPS * (size_t)c->n_channels, hipMemcpyDeviceToHost));
  return RDSP_OK;
}
extern "C" int rdsp_chain_get_mask(rdsp_chain_t *c, float *host_out) {
  NEED(c);
  memcpy(host_out, c->groups[0].mask_nat.data(), sizeof(float) * 2 * (size_t)c->N);
  return RDSP_OK;
}
extern "C" int rdsp_chain_get_fir_taps(rdsp_chain_t *c, float *host_out) {
  NEED(c);
  memcpy(host_out, c->fir_nat.data(), sizeof(float) * 256);
  return RDSP_OK;
}

/* ---- the sketch as shipped inside one chain (round 6) -----------------------------------------------------------------
 * A chain created as the bare CONV stage (decim 1, 44.1 kHz, RDSP_DEMOD_IQ, no mixer offset, unit gains, AGC / ALS /
 * spectral stage off: what loop() runs, INO:198) can take the reference's own pre-processor and engine in front of it:
 * rdsp_chain_process then is IQ -> AudioSDRpreProcessor::update -> AudioSDR::update -> doConvolutionalProcessing, and the
 * rdsp_sdr_* / rdsp_pre_* setters reach those objects (rdsp_engine_t, rdsp_preproc_t: the image's arithmetic, bit for bit)
 * instead of this build's stand-ins.  The engine's coefficient tables come from the host (rdsp_sdr_load_engine_tables). */
extern "C" int rdsp_sdr_set_engine_literal(rdsp_chain_t *c, int on) {
  NEED(c);
  if (!on) {
    if (c->engine) { rdsp_engine_destroy(c->engine); c->engine = nullptr; }
    if (c->pre) { rdsp_preproc_destroy(c->pre); c->pre = nullptr; }
    return RDSP_OK;
  }
  if (c->engine) return RDSP_OK;
  const rdsp_chain_config_t &cf = c->cfg;
  if (c->decim != 1 || cf.fs_in != 44100.0 || cf.demod != RDSP_DEMOD_IQ || cf.nco_hz != 0.0 || cf.agc_mode != RDSP_AGC_OFF ||
      cf.als_mode != RDSP_ALS_OFF || cf.spectral_nr != 0 || cf.input_gain != 1.0f || cf.output_gain != 1.0f || cf.iq_balance != 1.0f ||
      c->groups.size() != 1) {
    rdsp_set_error("rdsp_sdr_set_engine_literal: the chain must be the bare CONV stage (decim 1, 44.1 kHz, RDSP_DEMOD_IQ, nco 0, "
                   "AGC / ALS / spectral stage off, unit gains, one group)");
    return RDSP_ERR_INVALID;
  }
  if (check_device(c) != RDSP_OK) return RDSP_ERR_HIP;
  int rc = rdsp_preproc_create(c->n_channels, c->device, &c->pre);
  if (rc == RDSP_OK) rc = rdsp_engine_create(c->n_channels, c->device, c->max_blocks, &c->engine);
  if (rc == RDSP_OK && !c->d_engine_io &&
      hipMalloc((void **)&c->d_engine_io, (size_t)c->n_channels * (size_t)c->max_blocks * RDSP_BLOCK * 2 * sizeof(int16_t)) != hipSuccess)
    rc = RDSP_ERR_NOMEM;
  if (rc != RDSP_OK) (void)rdsp_sdr_set_engine_literal(c, 0);
  return rc;
}
extern "C" int rdsp_sdr_load_engine_tables(rdsp_chain_t *c, const float *biquad_sets15x20, const float *hilbert64) {
  NEED(c);
  if (!c->engine) { rdsp_set_error("rdsp_sdr_load_engine_tables: rdsp_sdr_set_engine_literal(chain, 1) first"); return RDSP_ERR_INVALID; }
  return rdsp_engine_load_tables(c->engine, biquad_sets15x20, hilbert64);
}
extern "C" rdsp_engine_t *rdsp_chain_engine(rdsp_chain_t *c) { return c ? c->engine : nullptr; }
extern "C" rdsp_preproc_t *rdsp_chain_preproc(rdsp_chain_t *c) { return c ? c->pre : nullptr; }
