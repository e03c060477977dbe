/*
 * rdsp_io.c -- recorded-IQ reader and audio writer: the roles
 * AudioInputI2S / AudioOutputI2S + the SGTL5000 codec play at either end of the
 * sketch's graph (RadioDSP_SDR_RX.ino:52,55,159-169), for a host that has files
 * or pipes instead of an I2S bus.  Plain C, no GPU code.
 *
 * Formats (both little-endian, as every SDR recorder writes them):
 *   RAW  interleaved int16 I,Q,I,Q,... one receiver channel per file, no header
 *   WAV  RIFF/WAVE, PCM (format tag 1 or WAVE_FORMAT_EXTENSIBLE with the PCM
 *        sub-format), 16 bit, 2 channels: left = I, right = Q -- the wiring of
 *        the sketch, codec left/right -> IQinput ports 0/1 (.ino:71-72)
 * Audio out: int16 L,R pairs (what Q_out_L / Q_out_R carry to audio_out,
 * RDSP_convolutional.h:344-349), RAW or WAV at the decimated rate.
 */
#include "rdsp_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

struct rdsp_iq_reader {
  FILE *f;
  int format;          /* RDSP_IO_RAW / RDSP_IO_WAV */
  double sample_rate;  /* 0 when the container does not say */
  int64_t frames;      /* IQ pairs in the data chunk, -1 unknown */
  int64_t pos;         /* pairs delivered so far */
  int owns;            /* close f on close */
};

struct rdsp_audio_writer {
  FILE *f;
  int format;
  int64_t frames;
  long riff_size_at, data_size_at; /* header fields patched on close */
};

static uint32_t le32(const unsigned char *p) {
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint16_t le16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
static void put32(unsigned char *p, uint32_t v) {
  p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); p[2] = (unsigned char)(v >> 16); p[3] = (unsigned char)(v >> 24);
}
static void put16(unsigned char *p, uint16_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); }

static int host_is_little_endian(void) {
  const uint16_t one = 1;
  return *(const unsigned char *)&one == 1;
}

/* walk the RIFF chunks up to "data"; leaves the file positioned on the samples */
static int parse_wav(rdsp_iq_reader_t *r) {
  unsigned char h[12];
  if (fread(h, 1, 12, r->f) != 12 || memcmp(h, "RIFF", 4) != 0 || memcmp(h + 8, "WAVE", 4) != 0) {
    rdsp_set_error("not a RIFF/WAVE file");
    return RDSP_ERR_INVALID;
  }
  int have_fmt = 0;
  for (;;) {
    unsigned char ch[8];
    if (fread(ch, 1, 8, r->f) != 8) {
      rdsp_set_error("WAV: no data chunk");
      return RDSP_ERR_INVALID;
    }
    const uint32_t size = le32(ch + 4);
    if (memcmp(ch, "fmt ", 4) == 0) {
      unsigned char f[40];
      const uint32_t take = size < sizeof(f) ? size : (uint32_t)sizeof(f);
      if (size < 16 || fread(f, 1, take, r->f) != take) {
        rdsp_set_error("WAV: short fmt chunk");
        return RDSP_ERR_INVALID;
      }
      uint16_t tag = le16(f);
      const uint16_t nch = le16(f + 2), bits = le16(f + 14);
      if (tag == 0xFFFE && take >= 26) tag = le16(f + 24); /* extensible: sub-format GUID starts with the tag */
      if (tag != 1 || nch != 2 || bits != 16) {
        rdsp_set_error("WAV: need PCM 16-bit stereo (I left, Q right); got tag %u, %u channels, %u bits", tag, nch, bits);
        return RDSP_ERR_UNSUPPORTED;
      }
      r->sample_rate = (double)le32(f + 4);
      have_fmt = 1;
      const long rest = (long)(size - take) + (long)(size & 1u);
      if (rest && fseek(r->f, rest, SEEK_CUR) != 0) return RDSP_ERR_INVALID;
    } else if (memcmp(ch, "data", 4) == 0) {
      if (!have_fmt) {
        rdsp_set_error("WAV: data before fmt");
        return RDSP_ERR_INVALID;
      }
      /* 0 and 0xFFFFFFFF are what recorders leave in streams they never finalised */
      r->frames = (size == 0u || size == 0xFFFFFFFFu) ? -1 : (int64_t)(size / 4u);
      return RDSP_OK;
    } else {
      if (fseek(r->f, (long)size + (long)(size & 1u), SEEK_CUR) != 0) {
        rdsp_set_error("WAV: cannot skip chunk");
        return RDSP_ERR_INVALID;
      }
    }
  }
}

int rdsp_iq_reader_open(const char *path, int format, rdsp_iq_reader_t **out) {
  if (!path || !out || format < RDSP_IO_AUTO || format > RDSP_IO_WAV) {
    rdsp_set_error("rdsp_iq_reader_open: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (!host_is_little_endian()) {
    rdsp_set_error("big-endian hosts are not supported by the file layer");
    return RDSP_ERR_UNSUPPORTED;
  }
  FILE *f = fopen(path, "rb");
  if (!f) {
    rdsp_set_error("cannot open %s", path);
    return RDSP_ERR_INVALID;
  }
  rdsp_iq_reader_t *r = (rdsp_iq_reader_t *)calloc(1, sizeof(*r));
  if (!r) { fclose(f); return RDSP_ERR_NOMEM; }
  r->f = f;
  r->owns = 1;
  r->frames = -1;
  if (format == RDSP_IO_AUTO) {
    unsigned char m[4] = {0, 0, 0, 0};
    const size_t got = fread(m, 1, 4, f);
    rewind(f);
    format = (got == 4 && memcmp(m, "RIFF", 4) == 0) ? RDSP_IO_WAV : RDSP_IO_RAW;
  }
  r->format = format;
  if (format == RDSP_IO_WAV) {
    const int rc = parse_wav(r);
    if (rc != RDSP_OK) { fclose(f); free(r); return rc; }
  } else if (fseek(f, 0, SEEK_END) == 0) {
    const long bytes = ftell(f);
    rewind(f);
    if (bytes >= 0) r->frames = (int64_t)(bytes / 4);
  }
  *out = r;
  return RDSP_OK;
}

double rdsp_iq_reader_sample_rate(const rdsp_iq_reader_t *r) { return r ? r->sample_rate : 0.0; }
int64_t rdsp_iq_reader_frames(const rdsp_iq_reader_t *r) { return r ? r->frames : -1; }
int rdsp_iq_reader_format(const rdsp_iq_reader_t *r) { return r ? r->format : 0; }

/* up to n IQ pairs into dst[2*n]; returns the pairs delivered (short only at the end) */
size_t rdsp_iq_reader_read(rdsp_iq_reader_t *r, int16_t *dst, size_t n) {
  if (!r || !dst) return 0;
  if (r->frames >= 0 && (int64_t)n > r->frames - r->pos) n = (size_t)(r->frames - r->pos);
  const size_t got = fread(dst, 4, n, r->f);
  r->pos += (int64_t)got;
  return got;
}

void rdsp_iq_reader_close(rdsp_iq_reader_t *r) {
  if (!r) return;
  if (r->owns && r->f) fclose(r->f);
  free(r);
}

int rdsp_audio_writer_open(const char *path, int format, double sample_rate, rdsp_audio_writer_t **out) {
  if (!path || !out || (format != RDSP_IO_RAW && format != RDSP_IO_WAV) || (format == RDSP_IO_WAV && sample_rate <= 0.0)) {
    rdsp_set_error("rdsp_audio_writer_open: bad argument");
    return RDSP_ERR_INVALID;
  }
  if (!host_is_little_endian()) {
    rdsp_set_error("big-endian hosts are not supported by the file layer");
    return RDSP_ERR_UNSUPPORTED;
  }
  FILE *f = fopen(path, "wb");
  if (!f) {
    rdsp_set_error("cannot create %s", path);
    return RDSP_ERR_INVALID;
  }
  rdsp_audio_writer_t *w = (rdsp_audio_writer_t *)calloc(1, sizeof(*w));
  if (!w) { fclose(f); return RDSP_ERR_NOMEM; }
  w->f = f;
  w->format = format;
  if (format == RDSP_IO_WAV) {
    unsigned char h[44];
    const uint32_t rate = (uint32_t)(sample_rate + 0.5);
    memcpy(h, "RIFF", 4); put32(h + 4, 36); memcpy(h + 8, "WAVEfmt ", 8); put32(h + 16, 16);
    put16(h + 20, 1); put16(h + 22, 2); put32(h + 24, rate); put32(h + 28, rate * 4u);
    put16(h + 32, 4); put16(h + 34, 16); memcpy(h + 36, "data", 4); put32(h + 40, 0);
    if (fwrite(h, 1, 44, f) != 44) { fclose(f); free(w); return RDSP_ERR_INVALID; }
    w->riff_size_at = 4;
    w->data_size_at = 40;
  }
  *out = w;
  return RDSP_OK;
}

/* n L,R pairs */
size_t rdsp_audio_writer_write(rdsp_audio_writer_t *w, const int16_t *lr, size_t n) {
  if (!w || !lr) return 0;
  const size_t put = fwrite(lr, 4, n, w->f);
  w->frames += (int64_t)put;
  return put;
}

int64_t rdsp_audio_writer_frames(const rdsp_audio_writer_t *w) { return w ? w->frames : 0; }

int rdsp_audio_writer_close(rdsp_audio_writer_t *w) {
  if (!w) return RDSP_ERR_INVALID;
  int rc = RDSP_OK;
  if (w->format == RDSP_IO_WAV) {
    const uint64_t bytes = (uint64_t)w->frames * 4u;
    const uint32_t data = bytes > 0xFFFFFFF0u ? 0xFFFFFFFFu : (uint32_t)bytes;
    unsigned char b[4];
    put32(b, data == 0xFFFFFFFFu ? data : data + 36u);
    if (fseek(w->f, w->riff_size_at, SEEK_SET) != 0 || fwrite(b, 1, 4, w->f) != 4) rc = RDSP_ERR_INVALID;
    put32(b, data);
    if (fseek(w->f, w->data_size_at, SEEK_SET) != 0 || fwrite(b, 1, 4, w->f) != 4) rc = RDSP_ERR_INVALID;
  }
  if (fclose(w->f) != 0) rc = RDSP_ERR_INVALID;
  free(w);
  return rc;
}


/* ---- I2S channel-slip estimate for a recording (INO:117, rdsp_pre_setIQslip) --------------------
 * A front end with the fault delivers one rail a sample behind the other.  On a stream with a dominant
 * one-sided line (a carrier, a tone: what a receiver's IQ stream normally holds) the misalignment shows
 * as a loss of image rejection: the line at +f leaks to -f by tan(pi f / fs).  The three hypotheses --
 * slip 0 (rails aligned), +1 (pair I[n-1] with Q[n]: what corrects a LATE Q rail), -1 (pair I[n] with
 * Q[n-1]: a late I rail) -- are applied in turn to the first 2^k samples (Hann window, one transform each)
 * and the rejection of the strongest line's image is measured under each.  A correction is only
 * recommended on evidence: the winner must reject the image by RDSP_SLIP_MIN_REJECTION_DB at least and beat
 * "no slip" by RDSP_SLIP_MARGIN_DB -- on noise, a real-valued or a double-side-band signal all three sit
 * near 0 dB, rounding would decide, and a spurious +-1 handed to rdsp_pre_setIQslip would destroy the image
 * rejection of a healthy recording.  Returns RDSP_OK and the value to pass to rdsp_pre_setIQslip in *slip
 * (0 without such evidence); rejection_db[3] (optional) gets the rejection under slip 0, +1, -1. */
#define RDSP_SLIP_MIN_REJECTION_DB 15.0
#define RDSP_SLIP_MARGIN_DB 10.0
int rdsp_estimate_iq_slip(const int16_t *iq, size_t n_samples, int *slip, double *rejection_db) {
  if (!iq || !slip || n_samples < 258) {
    rdsp_set_error("rdsp_estimate_iq_slip: bad argument (at least 258 samples)");
    return RDSP_ERR_INVALID;
  }
  int n = 256;
  while ((size_t)(2 * n) + 2 <= n_samples && n < 16384) n *= 2;
  double *re = (double *)malloc(sizeof(double) * 2 * (size_t)n);
  if (!re) {
    rdsp_set_error("rdsp_estimate_iq_slip: out of memory (%d-point transform)", n);
    return RDSP_ERR_NOMEM;
  }
  double *im = re + n;
  static const int hyp[3] = {0, 1, -1};
  double db[3] = {0.0, 0.0, 0.0};
  *slip = 0;
  for (int h = 0; h < 3; h++) {
    for (int i = 0; i < n; i++) { /* sample i + 1 of the recording, so that its predecessor exists */
      const double w = 0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * (double)i / (double)n);
      const int ii = (hyp[h] > 0) ? i : i + 1, qi = (hyp[h] < 0) ? i : i + 1;
      re[i] = w * (double)iq[2 * ii];
      im[i] = w * (double)iq[2 * qi + 1];
    }
    rdsp_host_fft(re, im, n);
    int kmax = 3;
    double pmax = -1.0;
    for (int k = 3; k < n - 2; k++) { /* strongest line away from DC */
      const double pw = re[k] * re[k] + im[k] * im[k];
      if (pw > pmax) { pmax = pw; kmax = k; }
    }
    double pimg = 0.0;
    for (int d = -2; d <= 2; d++) { /* its image, main lobe of the window included */
      const int k = ((n - kmax) + d + n) % n;
      const double pw = re[k] * re[k] + im[k] * im[k];
      if (pw > pimg) pimg = pw;
    }
    db[h] = 10.0 * log10(pmax / (pimg + 1e-30) + 1e-30);
    if (rejection_db) rejection_db[h] = db[h];
  }
  free(re);
  const int w = db[1] >= db[2] ? 1 : 2;
  if (db[w] >= RDSP_SLIP_MIN_REJECTION_DB && db[w] >= db[0] + RDSP_SLIP_MARGIN_DB) *slip = hyp[w];
  return RDSP_OK;
}
