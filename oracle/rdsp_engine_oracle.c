/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's un-vendored AudioSDR engine (`AudioSDR SDR;`,
 * /root/reference/src/RadioDSP_SDR_RX/RadioDSP_SDR_RX.ino:54, wired :81-86, configured :117-139, driven from
 * RDSP_controls.h:149-423).  Never linked by the product.
 *
 * The engine's source is not in the reference tree (Derek Rowell's AudioSDR library, no version pinned), but its
 * compiled code is: pre_compiled/RadioDSP_SDR_RX.ino.hex.  This file follows that code routine by routine, as read
 * with tests/golden/thumb_dis.py (ITCM addresses are given at each function), and is PINNED on it: the image's own
 * AudioSDR::update() runs under tests/golden/thumb_emu.py, tests/golden/make_engine_kat.py records its int16 output
 * and the float buffers between its stages for seeded inputs, and tests/test_engine_kat.py holds this restatement to
 * those bits (every stage, every case).  The order of operations, which products are rounded before they are added
 * and which are fused (VFMA), where the image widens to double and narrows again, its truncating conversions and its
 * comparisons are reproduced as found, quirks included; they are noted where they matter.
 *
 * Three tables the engine copies out of its initialised data are inputs here (tests/golden/firmware_tables.npz holds
 * them as data): fifteen sets of four biquad sections, 64 taps of one side of the Hilbert transformer -- neither has a
 * closed form -- and the 257-entry sine table, which is generated (sin(2 pi k / 256) to eight decimal places, the
 * literals of the library's header) and tested equal to the image's.
 *
 * Build: part of oracle/liboracle.so (oracle/Makefile), -ffp-contract=off; fused operations are explicit fmaf / fma. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rdsp_oracle.h"

#define TWO_PI_F 6.2831854820251465f /* the float the image holds for 2 pi */
#define K_RAD_PER_HZ 0.00014247586659621447f /* 2 pi / 44100 as the image's float */

static int32_t trunc_s32(double x) { /* VCVT.S32.F64: toward zero, saturating, NaN -> 0 */
  if (x != x) return 0;
  if (x >= 2147483647.0) return INT32_MAX;
  if (x <= -2147483648.0) return INT32_MIN;
  return (int32_t)x;
}
static float bits_f(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static uint32_t f_bits(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* ---- newlib's expf and frexpf as the image links them (setup time only: the AGC's gain curve) ---------------------
 * 0x13690 is the old wrapper (w_expf.c: _LIB_VERSION, matherr) around __ieee754_expf at 0x13ff0: Sun's e_expf.c,
 * restated here from its published algorithm (argument reduction by ln2 in two parts, a degree-5 polynomial in r^2,
 * scaling through the exponent field).  Pinned by tests/test_engine_kat.py against the image's routine on the
 * fixture's sample points. */
float orc_newlib_expf(float x) {
  static const float halF[2] = {0.5f, -0.5f}, huge = 1.0e+30f, twom100 = 7.8886090522e-31f, o_threshold = 8.8721679688e+01f,
                     u_threshold = -1.0397208405e+02f, ln2HI[2] = {6.9313812256e-01f, -6.9313812256e-01f},
                     ln2LO[2] = {9.0580006145e-06f, -9.0580006145e-06f}, invln2 = 1.4426950216e+00f, P1 = 1.6666667163e-01f,
                     P2 = -2.7777778450e-03f, P3 = 6.6137559770e-05f, P4 = -1.6533901999e-06f, P5 = 4.1381369442e-08f;
  float y, hi = 0.0f, lo = 0.0f, c, t;
  int32_t k = 0, xsb;
  uint32_t sx = f_bits(x), hx;
  xsb = (sx >> 31) & 1;
  hx = sx & 0x7fffffff;
  if (hx > 0x7f800000) return x + x;
  if (hx == 0x7f800000) return xsb == 0 ? x : 0.0f;
  if (x > o_threshold) return huge * huge;
  if (x < u_threshold) return twom100 * twom100;
  if (hx > 0x3eb17218) {        /* |x| > 0.5 ln2 */
    if (hx < 0x3F851592) {      /* and |x| < 1.5 ln2 */
      hi = x - ln2HI[xsb]; lo = ln2LO[xsb]; k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + halF[xsb]);
      t = (float)k;
      hi = x - t * ln2HI[0];
      lo = t * ln2LO[0];
    }
    x = hi - lo;
  } else if (hx < 0x31800000) { /* |x| < 2^-28 */
    if (huge + x > 1.0f) return 1.0f + x;
  }
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0f - ((x * c) / (c - 2.0f) - x);
  y = 1.0f - ((lo - (x * c) / (2.0f - c)) - hi);
  if (k >= -125) return bits_f(f_bits(y) + ((uint32_t)k << 23));
  return bits_f(f_bits(y) + ((uint32_t)(k + 100) << 23)) * twom100;
}

/* ---- the object -------------------------------------------------------------------------------------------------- */
enum { NTAP = 9 };
struct orc_engine {
  float sets[15][20];    /* +0x22a8: the fifteen coefficient sets, in the image's order */
  float hilbert[64];     /* +0x275c */
  float sine[257];       /* +0x285c */
  int16_t hilbert_len, hilbert_delay; /* +0x2758 = 257, +0x275a = 128 */
  float if_centre, ssb_band, cw_band; /* +32 6890, +36 3000, +40 1000 */
  float input_gain, gain_i, gain_q, iq_balance, output_gain, tuning_offset; /* +0x630 +0x634 +0x638 +0x63c +0x640 +0x650 */
  uint16_t mode; uint8_t mute;        /* +1620, +1622 */
  orc_biquad_t pre_i, pre_q, am_i, am_q, audio; /* +0x658 +0x664 +0x670 +0x67c +0x688 (states +1684 ...) */
  /* ALS filter, +2004 ... */
  int16_t als_taps, als_delay; float als_mu; float als_line[256]; float als_w[128]; uint8_t als_on, als_notch, als_adaptive;
  /* AGC, +0xdf0 ... +0x103d */
  float agc_attack_a, agc_decay_a, agc_attack_b, agc_decay_b; /* +0xdf0 +0xdf4 +0xdfc +0xe00 */
  float agc_gain;                                             /* +0xe08 */
  float agc_curve[130];                                       /* +0xe10; entry 129 shares its word with +0x1014 */
  float agc_knee_db, agc_makeup, agc_slope, agc_threshold_db; /* +0x1018 +0x101c +0x1020 +0x1028 */
  float agc_in, agc_env; int32_t agc_hang_time, agc_hang; uint8_t agc_active, agc_on; /* +0x102c +0x1030 +0x1034 +0x1038 +0x103c +0x103d */
  /* noise blanker, +0x1040 ... +0x2259 */
  float nb_i[384], nb_q[384], nb_mask[384]; float nb_keep, nb_new, nb_ratio, nb_last, nb_avg; int16_t nb_before, nb_after;
  uint8_t nb_on, nb_hit;
  /* synchronous AM, +0x225c ... +0x22a0 */
  float sam_keep, sam_new, sam_hz_per_rad, sam_phase, sam_lock_lo, sam_lock_hi, sam_hz, sam_wn, sam_zeta, sam_kd, sam_ko,
      sam_g1, sam_g2, sam_ga, sam_gb;
  uint8_t sam_locked;
  uint16_t audio_id; uint8_t audio_on; /* +0x22a4, +0x22a6 */
  /* statics of the library's translation unit */
  float nco_phase, am_phase, i_line[512], q_line[512], sam_cos, sam_sin, sam_u, sam_err;
  float I[128], Q[128], A[128];        /* +0x230, +0x430, +0x30 */
  float *tap;                          /* NTAP x 2 x 128, or null */
};

static void tap(orc_engine_t *e, int k, const float *a, const float *b) {
  if (!e->tap) return;
  memcpy(e->tap + (size_t)k * 256, a, 512);
  if (b) memcpy(e->tap + (size_t)k * 256 + 128, b, 512);
}

/* arm_biquad_cascade_df1_init_f32 (0x128b0): the instance points at the set, the state is cleared */
static void df1_init(orc_biquad_t *b, const float *coef20) { orc_biquad_init(b, 4, coef20); }

/* ---- AGC set-up: 0xdf14 (defaults), 0xdd40 (the gain curve), 0xdfe0 (setAGCmode) ----------------------------------- */
static void agc_curve(orc_engine_t *e) { /* 0xdd40 */
  const double ln10ish = 2.3025, db_per_octave = 6.026; /* the image's own 2.3025 and 6.026 (not ln 10 and 6.0206) */
  const double T = (double)e->agc_threshold_db, W = (double)e->agc_knee_db;
  const float x_lo = orc_newlib_expf((float)(((T - W * 0.5) * ln10ish) / 20.0));
  const float x_hi = orc_newlib_expf((float)(((T + W * 0.5) * ln10ish) / 20.0));
  for (int i = 0; i < 130; i++) {
    const float x = (float)i * 0.0078125f;
    int ex;
    const float m = frexpf(fabsf(x), &ex);
    if (x_lo > x) { e->agc_curve[i] = 1.0f; continue; }
    float p = fmaf(m, 1.2314958572387695f, -4.1185250282287598f);
    p = fmaf(m, p, 6.021970272064209f);
    p = fmaf(m, p, -3.1339645385742188f);
    const float log2x = p + (float)ex;
    const float xdb = (float)((double)log2x * db_per_octave);
    float gdb;
    if (x_hi >= x) { /* inside the knee (0xde90) */
      const double d = fma(W, 0.5, (double)(xdb - e->agc_threshold_db));
      const double q = ((((double)e->agc_slope - 1.0) * d) * d) / (W + W);
      gdb = (float)(q + (double)xdb) - xdb;
    } else {
      gdb = fmaf(xdb - e->agc_threshold_db, e->agc_slope, e->agc_threshold_db) - xdb;
    }
    e->agc_curve[i] = orc_newlib_expf((float)(((double)gdb * ln10ish) / 20.0));
  }
}
void orc_engine_setAGCmode(orc_engine_t *e, int mode) { /* 0xdfe0; constants as the image stores them (bit patterns) */
  switch (mode) {
    case 0: e->agc_on = 0; return;
    case 1: e->agc_attack_a = bits_f(0x3f79673b); e->agc_attack_b = bits_f(0x3cd318a0); e->agc_decay_a = bits_f(0x3f7fddca);
            e->agc_decay_b = bits_f(0x3a08d800); e->agc_hang_time = 4410; break;
    case 2: e->agc_attack_a = bits_f(0x3f7d5732); e->agc_attack_b = bits_f(0x3c2a3380); e->agc_decay_a = bits_f(0x3f7ff250);
            e->agc_decay_b = bits_f(0x395b0000); e->agc_hang_time = 22050; break;
    case 3: e->agc_attack_a = bits_f(0x3f7eaab6); e->agc_attack_b = bits_f(0x3baaa500); e->agc_decay_a = bits_f(0x3f7ff928);
            e->agc_decay_b = bits_f(0x38db0000); e->agc_hang_time = 88200; break;
    default: return;
  }
  e->agc_on = 1;
}
void orc_engine_enableAGC(orc_engine_t *e) { e->agc_on = 1; } /* 0xdfd4 */
static void agc_init(orc_engine_t *e) { /* 0xdf14 */
  e->agc_threshold_db = -60.0f; e->agc_slope = bits_f(0x3dcccccd); e->agc_knee_db = 2.0f;
  orc_engine_setAGCmode(e, 2);            /* the medium set, hang time excepted: */
  e->agc_decay_a = bits_f(0x3f7ff928); e->agc_decay_b = bits_f(0x38db0000); e->agc_hang_time = 4410;
  e->agc_on = 1;
  agc_curve(e);
  e->agc_in = e->agc_env = 0.0f; e->agc_hang = 0; /* +3556 ... are cleared too; nothing reads them */
}

/* ---- synchronous-AM loop constants, 0xed34 ------------------------------------------------------------------------ */
static void sam_constants(orc_engine_t *e) {
  const float k = 1.0f / (e->sam_kd * e->sam_ko);
  const double zeta = (double)e->sam_zeta, wn = (double)e->sam_wn, k4 = (double)k * 4.0;
  const double den = 1.0 / (zeta * 4.0) + zeta;
  e->sam_g1 = (float)((k4 * zeta * wn) / den);
  e->sam_g2 = (float)((k4 * wn * wn) / (den * den));
  e->sam_ga = e->sam_g1 + e->sam_g2;
  e->sam_gb = e->sam_g2;
}

/* ---- AudioSDR::setDemodMode, 0xd798: returns the tuning offset (INO:139, CTL:337-407) ----------------------------- */
float orc_engine_setDemodMode(orc_engine_t *e, int mode) {
  const float *set = 0;
  e->mode = (uint16_t)mode;
  switch (e->mode) {
    case 0: e->tuning_offset = (float)fma((double)e->ssb_band, 0.5, (double)e->if_centre); set = e->sets[12]; break;
    case 1: e->tuning_offset = (float)fma(-(double)e->ssb_band, 0.5, (double)e->if_centre); set = e->sets[12]; break;
    case 6: e->tuning_offset = (float)fma(-(double)e->ssb_band, 0.5, (double)e->if_centre); set = e->sets[11]; break;
    case 2: e->tuning_offset = (float)fma((double)e->cw_band, 0.5, (double)e->if_centre); set = e->sets[10]; break;
    case 3: e->tuning_offset = (float)fma(-(double)e->cw_band, 0.5, (double)e->if_centre); set = e->sets[10]; break;
    case 4: case 5: e->tuning_offset = e->if_centre; set = e->sets[14]; break;
    default: return e->tuning_offset;
  }
  df1_init(&e->pre_i, set);
  df1_init(&e->pre_q, set);
  return e->tuning_offset;
}
/* 0xd97c: ids 0 ... 9 pick a set, 10 switches the filter off; the id is stored whatever it is */
void orc_engine_setAudioFilter(orc_engine_t *e, int id) {
  static const int set_of_id[10] = {7, 8, 9, 0, 1, 2, 3, 4, 5, 6};
  if (id == 10) e->audio_on = 0;
  else if (id >= 0 && id < 10) df1_init(&e->audio, e->sets[set_of_id[id]]);
  e->audio_id = (uint16_t)id;
}
void orc_engine_enableAudioFilter(orc_engine_t *e) { e->audio_on = 1; }  /* 0xd970 */
void orc_engine_setInputGain(orc_engine_t *e, float g) {                  /* 0xd8a0 */
  if (g > 10.0f) g = 10.0f;
  else if (g < 0.0f) g = 0.0f;
  e->input_gain = g; e->gain_i = e->iq_balance * g; e->gain_q = g;
}
void orc_engine_setIQgainBalance(orc_engine_t *e, float b) { e->iq_balance = b; e->gain_i = b * e->input_gain; e->gain_q = e->input_gain; } /* 0xd8f0 */
void orc_engine_setOutputGain(orc_engine_t *e, float g) { e->output_gain = g; }  /* 0xd918 */
void orc_engine_setMute(orc_engine_t *e, int on) { e->mute = (uint8_t)on; }       /* 0xd924 (it also parks a gain nothing reads) */
void orc_engine_enableALSfilter(orc_engine_t *e) {                                /* 0xdb2c */
  e->als_on = 1;
  memset(e->als_line, 0, sizeof e->als_line); memset(e->als_w, 0, sizeof e->als_w);
}
void orc_engine_disableALSfilter(orc_engine_t *e) { e->als_on = 0; }      /* 0xdb14 */
void orc_engine_setALSfilterNotch(orc_engine_t *e) { e->als_notch = 1; }  /* 0xdb1c */
void orc_engine_setALSfilterPeak(orc_engine_t *e) { e->als_notch = 0; }   /* not in the image (the sketch never calls it): the other value of the same flag */
void orc_engine_setALSfilterAdaptive(orc_engine_t *e) { e->als_adaptive = 1; } /* 0xdb24 */
void orc_engine_disableNoiseBlanker(orc_engine_t *e) { e->nb_on = 0; }    /* 0xe380 */
void orc_engine_enableNoiseBlanker(orc_engine_t *e) { e->nb_on = 1; }     /* not in the image: the constructor's own default */
void orc_engine_set_tap(orc_engine_t *e, float *buf) { e->tap = buf; }
float orc_engine_scalar(const orc_engine_t *e, int which) {
  switch (which) {
    case 0: return e->nco_phase; case 1: return e->agc_gain; case 2: return e->agc_env; case 3: return (float)e->agc_hang;
    case 4: return (float)e->agc_active; case 5: return e->sam_hz; case 6: return (float)e->sam_locked; case 7: return e->am_phase;
    case 8: return e->nb_avg; case 9: return (float)e->nb_hit; case 10: return e->sam_phase;
    default: return 0.0f;
  }
}
const float *orc_engine_agc_curve(const orc_engine_t *e) { return e->agc_curve; }
const float *orc_engine_sine(const orc_engine_t *e) { return e->sine; }
const float *orc_engine_als_taps(const orc_engine_t *e) { return e->als_w; }

/* AudioSDR::AudioSDR 0x6744 + init 0xede4 */
orc_engine_t *orc_engine_create(const float *biquad_sets15x20, const float *hilbert64) {
  orc_engine_t *e = (orc_engine_t *)calloc(1, sizeof *e);
  if (!e) return 0;
  memcpy(e->sets, biquad_sets15x20, sizeof e->sets);
  memcpy(e->hilbert, hilbert64, sizeof e->hilbert);
  for (int k = 0; k < 257; k++) /* eight-place decimal literals */
    e->sine[k] = (float)(round(sin(2.0 * 3.14159265358979323846 * k / 256.0) * 1e8) / 1e8);
  e->hilbert_len = 257; e->hilbert_delay = 128;
  e->if_centre = 6890.0f; e->ssb_band = 3000.0f; e->cw_band = 1000.0f;
  e->input_gain = e->gain_i = e->gain_q = e->iq_balance = e->output_gain = 1.0f;
  e->als_taps = 55; e->als_delay = 3; e->als_mu = 0.5f; e->als_on = 0; e->als_notch = 1; e->als_adaptive = 1;
  e->agc_makeup = 10.0f; e->agc_active = 1; /* +0x103c: the constructor's value, until the AGC first runs */
  e->nb_keep = 0.995f; e->nb_new = bits_f(0x3ba3d700); e->nb_ratio = 1.2f; e->nb_avg = 10.0f; e->nb_before = e->nb_after = 10; e->nb_on = 1;
  e->sam_keep = 0.995f; e->sam_new = bits_f(0x3ba3d700); e->sam_hz_per_rad = bits_f(0x45db55dd); e->sam_lock_lo = 3890.0f;
  e->sam_lock_hi = 9890.0f; e->sam_hz = 1890.0f; e->sam_wn = bits_f(0x3e50fac7); e->sam_zeta = 2.0f; e->sam_kd = e->sam_ko = 1.0f;
  df1_init(&e->audio, e->sets[3]);
  df1_init(&e->pre_i, e->sets[12]); df1_init(&e->pre_q, e->sets[12]);
  df1_init(&e->am_i, e->sets[13]); df1_init(&e->am_q, e->sets[13]);
  agc_init(e);
  sam_constants(e);
  for (int i = 0; i < 384; i++) e->nb_mask[i] = 1.0f;
  orc_engine_setDemodMode(e, 0);
  e->audio_id = 0; /* the constructor never sets it */
  e->mute = 0;
  return e;
}
void orc_engine_destroy(orc_engine_t *e) { free(e); }

/* ---- table oscillator: cos and sin of a phase in [0, 2 pi) by linear interpolation in the 256-step table ------------ */
static float table_sin(const orc_engine_t *e, float ph) {
  const int32_t idx = trunc_s32(((double)ph * 65535.0) / (double)TWO_PI_F);
  const int hi = (idx >> 8) & 0xff;
  const float lo = (float)(uint32_t)(idx & 0xff);
  const float t0 = e->sine[hi], t1 = e->sine[hi + 1];
  return (float)fma((double)((t1 - t0) * lo), 0.00390625, (double)t0);
}
/* the frequency shifter: the loop at 0xe94e of update() and the routine 0xd600 are the same arithmetic */
static void freq_shift(const orc_engine_t *e, float *I, float *Q, float step, float *phase) {
  float ph = *phase;
  for (int i = 0; i < 128; i++) {
    float pc = (float)((double)ph + 1.5707963267948966);
    if (pc >= TWO_PI_F) pc -= TWO_PI_F;
    if (pc < 0.0f) pc += TWO_PI_F;
    const float c = table_sin(e, pc);
    float ps = ph >= TWO_PI_F ? ph - TWO_PI_F : ph;
    ph = ph + step;
    if (ps < 0.0f) ps += TWO_PI_F;
    const float s = table_sin(e, ps);
    const float x = I[i], y = Q[i];
    I[i] = fmaf(x, c, -(s * y));
    Q[i] = fmaf(y, c, x * s);
    if (ph > TWO_PI_F) ph -= TWO_PI_F;
    else if (ph < 0.0f) ph += TWO_PI_F;
  }
  *phase = ph;
}
static float quick_sqrt2(float p) { /* 0xec9c: a bit-pattern first guess and two Newton steps */
  const float g = bits_f((f_bits(p) >> 1) + 0x1fa00000u + 0x1b4000u + 3886u);
  float y = (p / g + g) * 0.5f;
  return (p / y + y) * 0.5f;
}
static float quick_sqrt1(float p) { /* 0xe27e: the same guess, one step */
  const float g = bits_f((f_bits(p) >> 1) + 0x1fa00000u + 0x1b4000u + 3886u);
  return (p / g + g) * 0.5f;
}

/* ---- impulse noise blanker, 0xe14c: two blocks of delay, a running average of |I + jQ|, blanking masks --------------- */
static void noise_blanker(orc_engine_t *e, float *I, float *Q) {
  static const float taper[7] = {0.933f, 0.75f, 0.5f, 0.25f, 0.067f, 0.0f, 0.0f}; /* 0x20003888 */
  e->nb_hit = 0;
  memmove(e->nb_i, e->nb_i + 128, 1024); memcpy(e->nb_i + 256, I, 512);
  memmove(e->nb_q, e->nb_q + 128, 1024); memcpy(e->nb_q + 256, Q, 512);
  memmove(e->nb_mask, e->nb_mask + 128, 1024);
  for (int i = 0; i < 128; i++) e->nb_mask[256 + i] = 1.0f;
  float avg = e->nb_avg, mag = 0.0f;
  for (int n = 78; n < 256; n++) {
    const float limit = avg * e->nb_ratio;
    mag = quick_sqrt1(fmaf(e->nb_i[n], e->nb_i[n], e->nb_q[n] * e->nb_q[n]));
    if (limit < mag) {
      if (-(int)e->nb_before <= (int)e->nb_after)
        for (int j = n - e->nb_before; j <= n + e->nb_after; j++) e->nb_mask[j] = 0.0f;
      e->nb_hit = 1;
    }
    avg = fmaf(avg, e->nb_keep, mag * e->nb_new);
  }
  e->nb_avg = avg;
  e->nb_last = mag;
  for (int i = 128; i < 256; i++) /* the middle block: a one-sided taper in front of every 0 -> 1 step */
    if (e->nb_mask[i] == 1.0f && e->nb_mask[i - 1] == 0.0f) memcpy(e->nb_mask + i - 7, taper, sizeof taper);
  for (int i = 0; i < 128; i++) { I[i] = e->nb_mask[i] * e->nb_i[i]; Q[i] = e->nb_mask[i] * e->nb_q[i]; }
}

/* ---- synchronous AM, 0xe390: a phase-locked loop on the IF signal, arctangent by a cubic ------------------------------ */
static void sam(orc_engine_t *e, float *I, float *Q) {
  const float HALF_PI = 1.5707963705062866f, A1 = 0.97239410877227783f, A3 = -0.19194795191287994f;
  float c = e->sam_cos, s = e->sam_sin, u_prev = e->sam_u, err_prev = e->sam_err, hz = e->sam_hz, ph = e->sam_phase;
  int locked = 0;
  float err = 0.0f, u = 0.0f;
  for (int i = 0; i < 128; i++) {
    const float q = Q[i], x = I[i];
    const float re = fmaf(x, c, q * s), im = fmaf(q, c, -(s * x));
    if (re == 0.0f) err = im > 0.0f ? HALF_PI : (im < 0.0f ? -HALF_PI : 0.0f);
    else if (fabsf(re) > fabsf(im)) {
      const float z = im / re;
      err = fmaf(z, z * A3, A1) * z;
      if (!(re > 0.0f)) err = (float)(im >= 0.0f ? (double)err + 3.1415926535897931 : (double)err - 3.1415926535897931);
    } else {
      const float z = re / im;
      err = fmaf(-z, fmaf(z, z * A3, A1), im > 0.0f ? HALF_PI : -HALF_PI);
    }
    u = fmaf(err, e->sam_ga, e->sam_gb * err_prev);
    const double phd = fma((double)(u + u_prev), 0.5, (double)ph);
    hz = fmaf(e->sam_keep, hz, (u * e->sam_hz_per_rad) * e->sam_new);
    ph = (float)phd;
    if ((double)ph >= 3.1415926535897931) ph -= TWO_PI_F;
    if ((double)ph < -3.1415926535897931) ph += TWO_PI_F;
    locked = hz > e->sam_lock_lo ? (hz < e->sam_lock_hi) : 0;
    float pc = (float)((double)ph + 1.5707963267948966);
    if (pc >= TWO_PI_F) pc -= TWO_PI_F;
    if (pc < 0.0f) pc += TWO_PI_F;
    c = table_sin(e, pc);
    float ps = ph >= TWO_PI_F ? ph - TWO_PI_F : ph;
    if (ps < 0.0f) ps += TWO_PI_F;
    s = table_sin(e, ps);
    if (locked) {
      const float xi = I[i], xq = Q[i];
      I[i] = fmaf(xi, c, xq * s);
      Q[i] = fmaf(-xi, s, xq * c);
    }
    u_prev = u; err_prev = err;
  }
  e->sam_hz = hz; e->sam_phase = ph; e->sam_err = err; e->sam_cos = c; e->sam_u = u; e->sam_sin = s; e->sam_locked = (uint8_t)locked;
}

/* ---- AGC, 0xdb58: peak envelope with attack / hang / decay, gain by a table over the envelope ------------------------- */
static float agc_lookup(const orc_engine_t *e, float env) {
  const int32_t idx = trunc_s32((double)env * 32767.0);
  int hi = (idx >> 8) & 0xff, hi1;
  if (hi > 127) { hi = 127; hi1 = 128; } else hi1 = hi + 1;
  const float frac = (float)(uint32_t)(idx & 0xff) * 0.00390625f;
  const float t0 = e->agc_curve[hi];
  return fmaf(frac, e->agc_curve[hi1] - t0, t0);
}
static void agc(orc_engine_t *e, float *a) {
  int active = 0;
  for (int i = 0; i < 128; i++) {
    float in = fabsf(a[i]);
    if (in > 1.0f) in = 1.0f;
    e->agc_in = in;
    const float env = e->agc_env;
    float g;
    if (env < in) {
      e->agc_in = e->agc_env = fmaf(env, e->agc_attack_a, in * e->agc_attack_b);
      e->agc_hang = e->agc_hang_time;
      g = e->agc_gain = agc_lookup(e, e->agc_env);
    } else if (e->agc_hang == 0) {
      e->agc_in = e->agc_env = fmaf(env, e->agc_decay_a, in * e->agc_decay_b);
      g = e->agc_gain = agc_lookup(e, e->agc_env);
    } else {
      e->agc_hang--;
      g = e->agc_gain;
    }
    active = (double)g < 0.98999999999999999;
    float y = (g * e->agc_makeup) * a[i];
    if (y > 1.0f) y = 1.0f;
    else if (y < -1.0f) y = -1.0f;
    a[i] = y;
  }
  e->agc_active = (uint8_t)active;
}

/* ---- ALS filter, 0xda24: a delayed-input LMS line enhancer; taps move on every fourth sample --------------------------- */
static void als(orc_engine_t *e, float *a) {
  memcpy(e->als_line, e->als_line + 128, 512);
  memcpy(e->als_line + 128, a, 512);
  int cnt = 0;
  for (int n = 128; n < 256; n++) {
    float y = 0.0f;
    for (int k = 0; k < e->als_taps; k++) y = fmaf(e->als_w[k], e->als_line[n - e->als_delay - k], y);
    const float err = e->als_line[n] - y;
    if (e->als_adaptive) {
      if (cnt == 0)
        for (int k = 0; k < e->als_taps; k++) e->als_w[k] = fmaf(err * e->als_line[n - e->als_delay - k], e->als_mu, e->als_w[k]);
      cnt = (cnt + 1) & 3;
    }
    a[n - 128] = e->als_notch ? err : y;
  }
}

/* ---- AudioSDR::update, 0xe730 ---------------------------------------------------------------------------------------- */
void orc_engine_update(orc_engine_t *e, const int16_t *i128, const int16_t *q128, int16_t *out128) {
  float *I = e->I, *Q = e->Q, *A = e->A;
  const double gi = (double)e->gain_i, gq = (double)e->gain_q;
  for (int i = 0; i < 128; i++) { /* 0xe7b4: divided by 32767, in double */
    I[i] = (float)(((double)i128[i] / 32767.0) * gi);
    Q[i] = (float)(((double)q128[i] / 32767.0) * gq);
  }
  tap(e, 0, I, Q);
  if (e->nb_on) noise_blanker(e, I, Q);
  tap(e, 1, I, Q);
  orc_biquad_run(&e->pre_i, I, 128);
  orc_biquad_run(&e->pre_q, Q, 128);
  tap(e, 2, I, Q);
  const unsigned mode = e->mode;
  if (mode <= 3 || mode == 6) {
    freq_shift(e, I, Q, -(e->tuning_offset * K_RAD_PER_HZ), &e->nco_phase);
    tap(e, 3, I, Q);
    /* 0xea7e: both rails into 512-sample lines; I comes out delayed by 128, Q through the 257-tap Hilbert transformer
     * (odd taps only, antisymmetric: 64 products per sample) */
    memmove(e->i_line, e->i_line + 128, 1536); memcpy(e->i_line + 384, I, 512);
    memmove(e->q_line, e->q_line + 128, 1536); memcpy(e->q_line + 384, Q, 512);
    const int quarter = (e->hilbert_len < 0 ? e->hilbert_len + 3 : e->hilbert_len) >> 2;
    for (int n = 384; n < 512; n++) {
      float acc = 0.0f;
      for (int k = 0; k < quarter; k++) acc = fmaf(e->hilbert[k], e->q_line[n - 1 - 2 * k] - e->q_line[n - e->hilbert_len + 2 + 2 * k], acc);
      Q[n - 384] = acc;
      I[n - 384] = e->i_line[n - e->hilbert_delay];
    }
    tap(e, 4, I, Q);
    const int minus = mode == 6 || (mode & ~2u) == 1;
    for (int i = 0; i < 128; i++) A[i] = minus ? I[i] - Q[i] : I[i] + Q[i];
  } else if (mode == 4 || mode == 5) {
    orc_biquad_run(&e->pre_i, I, 128); /* 0xec1c: the IF filter a second time */
    orc_biquad_run(&e->pre_q, Q, 128);
    int envelope = mode == 4;
    if (mode == 5) {
      sam(e, I, Q);
      memcpy(A, I, 512);
      envelope = !e->sam_locked; /* 0xed02: out of lock, the envelope detector takes over */
    }
    if (envelope) {
      freq_shift(e, I, Q, -e->if_centre * K_RAD_PER_HZ, &e->am_phase);
      orc_biquad_run(&e->am_i, I, 128);
      orc_biquad_run(&e->am_q, Q, 128);
      for (int i = 0; i < 128; i++) A[i] = quick_sqrt2(fmaf(I[i], I[i], Q[i] * Q[i]));
    }
  }
  tap(e, 5, A, 0);
  if (e->audio_on) orc_biquad_run(&e->audio, A, 128);
  tap(e, 6, A, 0);
  if (e->agc_on) agc(e, A);
  tap(e, 7, A, 0);
  if (e->als_on) als(e, A);
  tap(e, 8, A, 0);
  for (int i = 0; i < 128; i++) /* 0xebfa: times 32767, toward zero, the low half-word stored */
    out128[i] = e->mute ? 0 : (int16_t)(uint16_t)trunc_s32((double)(A[i] * e->output_gain) * 32767.0);
}

/* ---- AudioSDRpreProcessor (`AudioSDRpreProcessor preProcessor;`, INO:53, wired INO:71-72, :117-118) ----------------------
 * ::update at ITCM 0xee88, ::startAutoI2SerrorDetection at 0xf084; the object is set up inline by the sketch's static
 * initialiser (0x93e8 ... 0x9442).  It repairs the one-sample slip between the I and Q rails that the Teensy's I2S input
 * can start up with: while detection is on, each block goes through a 128-point complex FFT; if the strongest bin
 * (5 ... 122) stands more than 10 x above the mean and its mirror image is less than 20 dB down, a bad-count rises, and at
 * the eleventh bad block in a row the correction moves on (none -> I delayed -> Q delayed -> none); after 1000 checked
 * blocks detection switches itself off.  The FFT here is the oracle's arm_cfft_f32 (5e-7 from the image's,
 * tests/test_firmware_kat.py): the decisions compare ratios against 10, so a last-bit difference matters only on a
 * knife edge; the int16 blocks it hands on are exact given the same decisions. */
struct orc_preproc { int16_t slip, saved, bad, checks; uint8_t swap, detect; float buf[256]; };
orc_preproc_t *orc_preproc_create(void) { return (orc_preproc_t *)calloc(1, sizeof(orc_preproc_t)); }
void orc_preproc_destroy(orc_preproc_t *p) { free(p); }
void orc_preproc_startAutoI2SerrorDetection(orc_preproc_t *p) { p->slip = 0; p->detect = 1; p->bad = 0; p->checks = 0; } /* 0xf084 */
void orc_preproc_swapIQ(orc_preproc_t *p, int on) { p->swap = (uint8_t)(on != 0); } /* INO:118; not in the image: the flag at +1072 */
int orc_preproc_state(const orc_preproc_t *p, int which) { return which == 0 ? p->slip : which == 1 ? p->bad : which == 2 ? p->checks : p->detect; }
void orc_preproc_update(orc_preproc_t *p, int16_t *i128, int16_t *q128) {
  if (p->slip == 1 || p->slip == -1) { /* 0xefd4 / 0xf040: one rail a sample late, the last sample carried to the next block */
    int16_t *d = p->slip == 1 ? i128 : q128;
    const int16_t last = d[127];
    memmove(d + 1, d, 127 * sizeof(int16_t));
    i128[0] = p->saved; /* as compiled (0xefe8 stores through the I block's pointer in both cases): with Q delayed, the carried
                         * Q sample lands in I[0] and Q[0] keeps the block's own first sample */
    p->saved = last;
  }
  if (p->detect) {
    for (int k = 0; k < 128; k++) { p->buf[2 * k] = (float)i128[k] / 32767.0f; p->buf[2 * k + 1] = (float)q128[k] / 32767.0f; }
    orc_cfft_f32(p->buf, 128, 0);
    orc_cmplx_mag_f32(p->buf, p->buf, 128);
    float top = 0.0f, sum = 0.0f;
    int at = 0;
    for (int k = 5; k < 123; k++) {
      sum = sum + p->buf[k];
      if (p->buf[k] > top) { top = p->buf[k]; at = k; }
    }
    const float mean = sum / 118.0f;
    int checks = p->checks;
    if ((double)top > (double)mean * 10.0) {
      const float image = p->buf[128 - at];
      if (top / image < 10.0f) {
        p->bad = (int16_t)(p->bad + 1);
        if (p->bad > 10) {
          int s = (int16_t)(p->slip + 1);
          p->bad = 0;
          if (s > 1) s = -1;
          p->slip = (int16_t)s;
          checks = 1;
        } else checks = (int16_t)(checks + 1);
      } else {
        checks = (int16_t)(checks + 1);
        p->bad = 0;
      }
      p->checks = (int16_t)checks;
    }
    if (checks > 1000) p->detect = 0;
  }
  if (p->swap)
    for (int k = 0; k < 128; k++) { const int16_t t = i128[k]; i128[k] = q128[k]; q128[k] = t; }
}
