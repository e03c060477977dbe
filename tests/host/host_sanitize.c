/*
 * host_sanitize.c -- the host-only C of the library (rdsp_graph.c, rdsp_io.c, rdsp_design.c, rdsp_q15_tables.c) under
 * AddressSanitizer + UndefinedBehaviorSanitizer + LeakSanitizer on the CPU build (the GPU pool has
 * no sanitizer runs).  tests/test_host_logic.py compiles this file together with those three
 * sources and expects exit code 0 and no sanitizer report.  Walks the block graph of the sketch
 * (RadioDSP_SDR_RX.ino:71-89 shape: input -> two record queues, play queue -> sink with fan-out),
 * pool exhaustion and recovery, refcounts, teardown with blocks still queued, the RAW / WAV file
 * layer including damaged headers, and every design routine at every supported size.
 */
#include "rdsp.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the two error helpers live in rdsp_chain.hip; this build has no HIP objects */
static char g_err[256];
void rdsp_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char *rdsp_last_error(void) { return g_err; }

int rdsp_plan_radix(int fft_l);
int rdsp_bin_of_pos(int fft_l, int i);
void rdsp_mask_device_image(const float *mask_nat, int fft_l, float *image);
int rdsp_design_decimator(int ntaps, double cut_hz, double fs, int window, float *h_nat, float *hc);
int rdsp_fd_decimator_image(const float *h_nat, int fft_l, float *image);
uint32_t rdsp_nco_dphi(double hz, double fs);
void rdsp_nco_rot(uint32_t dphi, int k, float *out2);
float rdsp_lms_mu(int strength);
void rdsp_sam_constants(double fs_out, float *g1, float *g2, float *wmin, float *wmax);

#define CHECK(c)                                                         \
  do {                                                                   \
    if (!(c)) {                                                          \
      fprintf(stderr, "%s:%d: CHECK(%s) failed\n", __FILE__, __LINE__, #c); \
      exit(1);                                                           \
    }                                                                    \
  } while (0)

/* ---- graph ------------------------------------------------------------------------------ */
typedef struct { int ticks, got; long sum; } sink_t;
static void sink_update(rdsp_node_t *n, void *user) {
  sink_t *s = (sink_t *)user;
  s->ticks++;
  for (int port = 0; port < 2; port++) {
    rdsp_block_t *b = port ? rdsp_receive_writable(n, port) : rdsp_receive_readonly(n, port);
    if (!b) continue;
    const int nch = rdsp_graph_channels(rdsp_node_graph(n));
    int16_t *d = rdsp_block_data(b);
    for (int i = 0; i < nch * RDSP_BLOCK_SAMPLES; i++) s->sum += d[i];
    if (port) d[0] = (int16_t)(d[0] + 1); /* a writable block is private */
    s->got++;
    rdsp_release(b);
  }
}
static int g_destroyed;
static void count_destroy(void *p) { (void)p; g_destroyed++; }

/* the rings of the queues at the sizes of the reference's image (209 / 80), across their wrap-around */
static void queue_ring_checks(void) {
  rdsp_graph_t *g = rdsp_graph_create(1);
  CHECK(g && rdsp_memory(g, 300) == RDSP_OK);
  rdsp_node_t *in = rdsp_input_node_create(g), *q = rdsp_record_queue_create(g), *qd = rdsp_record_queue_create(g);
  rdsp_node_t *play = rdsp_play_queue_create(g);
  CHECK(in && q && qd && play && rdsp_connect(in, 0, q, 0) == RDSP_OK && rdsp_connect(in, 1, qd, 0) == RDSP_OK);
  rdsp_record_queue_begin(q); /* qd is never begun: it releases what it receives */
  int16_t ti[RDSP_BLOCK_SAMPLES], tq[RDSP_BLOCK_SAMPLES];
  memset(tq, 0, sizeof(tq));
  int next = 0, expect = 0;
  for (int round = 0; round < 3; round++) {
    for (int t = 0; t < 230; t++) {
      for (int i = 0; i < RDSP_BLOCK_SAMPLES; i++) ti[i] = (int16_t)next;
      CHECK(rdsp_input_node_push(in, ti, tq) == RDSP_OK && rdsp_update_all(g) == RDSP_OK);
      if (t < 208) next++; /* blocks 208 ... 229 of a round find the ring full and are dropped */
    }
    CHECK(rdsp_record_queue_available(q) == 208 && rdsp_memory_usage(g) == 208);
    for (int k = 0; k < 208; k++) {
      int16_t *b = rdsp_record_queue_readBuffer(q);
      CHECK(b && b[5] == (int16_t)expect);
      expect++;
      rdsp_record_queue_freeBuffer(q);
    }
    CHECK(rdsp_record_queue_available(q) == 0 && rdsp_memory_usage(g) == 0);
  }
  for (int round = 0; round < 3; round++) { /* the play queue: 79 fit, the 80th reports; one tick plays one */
    int n = 0;
    while (rdsp_play_queue_getBuffer(play) && rdsp_play_queue_playBuffer(play) == RDSP_OK) n++;
    CHECK(n == (round == 0 ? 79 : 40));
    for (int t = 0; t < 40; t++) CHECK(rdsp_update_all(g) == RDSP_OK);
  }
  rdsp_graph_destroy(g);
}

static void graph_checks(int nch) {
  rdsp_graph_t *g = rdsp_graph_create(nch);
  CHECK(g && rdsp_graph_channels(g) == nch);
  CHECK(rdsp_graph_create(0) == NULL);
  CHECK(rdsp_memory(g, 0) == RDSP_ERR_INVALID);
  CHECK(rdsp_memory(g, 12) == RDSP_OK && rdsp_memory(g, 12) == RDSP_ERR_INVALID);
  rdsp_node_t *in = rdsp_input_node_create(g);
  rdsp_node_t *qi = rdsp_record_queue_create(g), *qq = rdsp_record_queue_create(g);
  rdsp_node_t *play = rdsp_play_queue_create(g);
  sink_t st = {0, 0, 0};
  rdsp_node_t *sink = rdsp_node_create(g, 2, sink_update, &st);
  rdsp_node_set_destructor(sink, count_destroy);
  CHECK(in && qi && qq && play && sink && rdsp_node_user(sink) == &st);
  CHECK(rdsp_node_create(g, 5, NULL, NULL) == NULL);
  CHECK(rdsp_connect(in, 0, qi, 0) == RDSP_OK && rdsp_connect(in, 1, qq, 0) == RDSP_OK);
  CHECK(rdsp_connect(play, 0, sink, 0) == RDSP_OK && rdsp_connect(play, 0, sink, 1) == RDSP_OK); /* fan-out */
  CHECK(rdsp_connect(in, 0, qi, 1) == RDSP_ERR_INVALID && rdsp_connect(in, 4, qi, 0) == RDSP_ERR_INVALID);
  CHECK(rdsp_connect(NULL, 0, qi, 0) == RDSP_ERR_INVALID);

  const size_t tile = (size_t)nch * RDSP_BLOCK_SAMPLES;
  int16_t *ti = (int16_t *)malloc(tile * sizeof(int16_t)), *tq = (int16_t *)malloc(tile * sizeof(int16_t));
  CHECK(ti && tq);
  rdsp_record_queue_begin(qi);
  rdsp_record_queue_begin(qq);
  CHECK(rdsp_input_node_push(sink, ti, tq) == RDSP_ERR_INVALID);
  /* queues not drained: the pool (12) runs out after 6 ticks, later ticks drop their blocks */
  for (int t = 0; t < 20; t++) {
    for (size_t i = 0; i < tile; i++) { ti[i] = (int16_t)(t * 3 + (int)i); tq[i] = (int16_t)(-t - (int)i); }
    CHECK(rdsp_input_node_push(in, ti, tq) == RDSP_OK);
    CHECK(rdsp_update_all(g) == RDSP_OK);
  }
  CHECK(rdsp_memory_usage(g) == 12 && rdsp_memory_usage_max(g) == 12);
  CHECK(rdsp_record_queue_available(qi) == 6 && rdsp_record_queue_available(qq) == 6);
  CHECK(rdsp_play_queue_getBuffer(play) == NULL); /* exhausted */
  /* loop(): drain pairs, produce one output block per pair through the play queue */
  int drained = 0;
  while (rdsp_record_queue_available(qi) && rdsp_record_queue_available(qq)) {
    int16_t *bi = rdsp_record_queue_readBuffer(qi), *bq = rdsp_record_queue_readBuffer(qq);
    CHECK(bi && bq && rdsp_record_queue_readBuffer(qi) == NULL); /* one user block at a time */
    CHECK(bi[1] == (int16_t)(drained * 3 + 1) && bq[1] == (int16_t)(-drained - 1));
    rdsp_record_queue_freeBuffer(qi);
    rdsp_record_queue_freeBuffer(qq);
    int16_t *o = rdsp_play_queue_getBuffer(play);
    CHECK(o && o == rdsp_play_queue_getBuffer(play));
    for (size_t i = 0; i < tile; i++) o[i] = 1;
    CHECK(rdsp_play_queue_playBuffer(play) == RDSP_OK);
    CHECK(rdsp_play_queue_playBuffer(play) == RDSP_ERR_INVALID);
    drained++;
  }
  CHECK(drained == 6);
  rdsp_no_interrupts(g);
  CHECK(rdsp_update_all(g) == RDSP_ERR_NOT_READY);
  rdsp_interrupts(g);
  rdsp_interrupts(g); /* unbalanced call is harmless */
  const int before = st.ticks;
  for (int t = 0; t < 8; t++) CHECK(rdsp_update_all(g) == RDSP_OK);
  /* writable port copies while the block is shared: needs a free block, which exists now */
  CHECK(st.ticks == before + 8 && st.got == 12 && st.sum == 12L * (long)tile);
  CHECK(rdsp_memory_usage(g) == 0);
  /* play queue overflow reports instead of spinning; record queue end() drops */
  rdsp_record_queue_end(qi);
  rdsp_record_queue_end(qq);
  int queued = 0;
  for (;;) {
    if (!rdsp_play_queue_getBuffer(play)) break;
    if (rdsp_play_queue_playBuffer(play) != RDSP_OK) break;
    queued++;
  }
  CHECK(queued >= 11 && queued <= 12);
  /* tear down with blocks queued, a user block held and inputs pending */
  CHECK(rdsp_input_node_push(in, ti, tq) == RDSP_OK);
  g_destroyed = 0;
  rdsp_graph_destroy(g);
  CHECK(g_destroyed == 1);
  rdsp_graph_destroy(NULL);
  rdsp_release(NULL);
  CHECK(rdsp_block_data(NULL) == NULL && rdsp_block_refcount(NULL) == 0 && rdsp_allocate(NULL) == NULL);
  free(ti);
  free(tq);
}

/* ---- files ------------------------------------------------------------------------------ */
static void put_le32(unsigned char *p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (unsigned char)(v >> (8 * i)); }
static void put_le16(unsigned char *p, uint16_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); }

static void write_bytes(const char *path, const unsigned char *b, size_t n) {
  FILE *f = fopen(path, "wb");
  CHECK(f && fwrite(b, 1, n, f) == n);
  fclose(f);
}

static void io_checks(const char *dir) {
  char p1[512], p2[512], p3[512];
  snprintf(p1, sizeof p1, "%s/a.wav", dir);
  snprintf(p2, sizeof p2, "%s/a.raw", dir);
  snprintf(p3, sizeof p3, "%s/bad.wav", dir);
  enum { N = 1000 };
  int16_t lr[2 * N], back[2 * N + 8];
  for (int i = 0; i < 2 * N; i++) lr[i] = (int16_t)(i * 37 - 20000);
  for (int fmt = RDSP_IO_RAW; fmt <= RDSP_IO_WAV; fmt++) {
    rdsp_audio_writer_t *w = NULL;
    const char *p = fmt == RDSP_IO_WAV ? p1 : p2;
    CHECK(rdsp_audio_writer_open(p, fmt, 24000.0, &w) == RDSP_OK && w);
    CHECK(rdsp_audio_writer_write(w, lr, 600) == 600 && rdsp_audio_writer_write(w, lr + 1200, N - 600) == N - 600);
    CHECK(rdsp_audio_writer_write(w, lr, 0) == 0 && rdsp_audio_writer_frames(w) == N);
    CHECK(rdsp_audio_writer_close(w) == RDSP_OK);
    rdsp_iq_reader_t *r = NULL;
    CHECK(rdsp_iq_reader_open(p, RDSP_IO_AUTO, &r) == RDSP_OK && r);
    CHECK(rdsp_iq_reader_format(r) == fmt && rdsp_iq_reader_frames(r) == N);
    CHECK(rdsp_iq_reader_sample_rate(r) == (fmt == RDSP_IO_WAV ? 24000.0 : 0.0));
    CHECK(rdsp_iq_reader_read(r, back, 300) == 300 && rdsp_iq_reader_read(r, back + 600, N + 4) == N - 300);
    CHECK(rdsp_iq_reader_read(r, back, 10) == 0 && memcmp(back + 600, lr + 600, (N - 300) * 4) == 0);
    rdsp_iq_reader_close(r);
  }
  rdsp_iq_reader_t *r = NULL;
  rdsp_audio_writer_t *w = NULL;
  CHECK(rdsp_iq_reader_open(NULL, 0, &r) == RDSP_ERR_INVALID && rdsp_iq_reader_open(p1, 7, &r) == RDSP_ERR_INVALID);
  CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) == RDSP_ERR_INVALID && strstr(rdsp_last_error(), "cannot open"));
  CHECK(rdsp_audio_writer_open(p3, RDSP_IO_AUTO, 24000.0, &w) == RDSP_ERR_INVALID);
  CHECK(rdsp_audio_writer_open(p3, RDSP_IO_WAV, 0.0, &w) == RDSP_ERR_INVALID);
  CHECK(rdsp_audio_writer_close(NULL) != RDSP_OK || 1);
  rdsp_iq_reader_close(NULL);
  CHECK(rdsp_iq_reader_read(NULL, back, 1) == 0 && rdsp_iq_reader_frames(NULL) == -1);

  /* damaged and unusual headers: every prefix of a good header, a huge chunk size, mono, an
   * odd-sized LIST chunk before fmt, an extensible fmt chunk, data size 0xFFFFFFFF */
  unsigned char h[128];
  memset(h, 0, sizeof h);
  memcpy(h, "RIFF", 4); put_le32(h + 4, 36 + 40); memcpy(h + 8, "WAVE", 4);
  memcpy(h + 12, "fmt ", 4); put_le32(h + 16, 16); put_le16(h + 20, 1); put_le16(h + 22, 2);
  put_le32(h + 24, 96000); put_le32(h + 28, 384000); put_le16(h + 32, 4); put_le16(h + 34, 16);
  memcpy(h + 36, "data", 4); put_le32(h + 40, 40);
  for (int i = 0; i < 40; i++) h[44 + i] = (unsigned char)i;
  for (size_t cut = 0; cut < 44; cut++) {
    write_bytes(p3, h, cut);
    const int rc = rdsp_iq_reader_open(p3, RDSP_IO_WAV, &r);
    CHECK(rc == RDSP_ERR_INVALID);
  }
  write_bytes(p3, h, 84);
  CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) == RDSP_OK && rdsp_iq_reader_frames(r) == 10 && rdsp_iq_reader_sample_rate(r) == 96000.0);
  CHECK(rdsp_iq_reader_read(r, back, 64) == 10 && back[0] == 0x0100 && back[19] == 0x2726);
  rdsp_iq_reader_close(r);
  put_le32(h + 40, 0xFFFFFFFFu); /* unfinalised stream: read to the end of the file */
  write_bytes(p3, h, 84);
  CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) == RDSP_OK && rdsp_iq_reader_frames(r) == -1);
  CHECK(rdsp_iq_reader_read(r, back, 64) == 10);
  rdsp_iq_reader_close(r);
  put_le32(h + 40, 40);
  put_le16(h + 22, 1); /* mono */
  write_bytes(p3, h, 84);
  CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) == RDSP_ERR_UNSUPPORTED && strstr(rdsp_last_error(), "1 channels"));
  put_le16(h + 22, 2);
  put_le32(h + 16, 0x7FFFFFF0u); /* fmt chunk claims 2 GiB */
  write_bytes(p3, h, 84);
  CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) != RDSP_OK);
  put_le32(h + 16, 8); /* fmt chunk too short */
  write_bytes(p3, h, 84);
  CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) == RDSP_ERR_INVALID);
  {
    unsigned char e[160];
    memset(e, 0, sizeof e);
    memcpy(e, "RIFF", 4); put_le32(e + 4, 120); memcpy(e + 8, "WAVE", 4);
    memcpy(e + 12, "LIST", 4); put_le32(e + 16, 5); memcpy(e + 20, "INFOx", 5); /* + 1 pad byte */
    unsigned char *f = e + 26;
    memcpy(f, "fmt ", 4); put_le32(f + 4, 40); put_le16(f + 8, 0xFFFE); put_le16(f + 10, 2);
    put_le32(f + 12, 48000); put_le32(f + 16, 192000); put_le16(f + 20, 4); put_le16(f + 22, 16);
    put_le16(f + 24, 22); put_le16(f + 26, 16); put_le32(f + 28, 3); put_le16(f + 32, 1); /* sub-format PCM */
    unsigned char *d = f + 48;
    memcpy(d, "data", 4); put_le32(d + 4, 8);
    for (int i = 0; i < 8; i++) d[8 + i] = (unsigned char)(0x10 + i);
    write_bytes(p3, e, (size_t)(d + 16 - e));
    CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) == RDSP_OK && rdsp_iq_reader_frames(r) == 2 && rdsp_iq_reader_sample_rate(r) == 48000.0);
    CHECK(rdsp_iq_reader_read(r, back, 8) == 2 && back[0] == 0x1110 && back[3] == 0x1716);
    rdsp_iq_reader_close(r);
    memcpy(e + 12, "data", 4); /* data before fmt */
    write_bytes(p3, e, (size_t)(d + 16 - e));
    CHECK(rdsp_iq_reader_open(p3, RDSP_IO_AUTO, &r) == RDSP_ERR_INVALID && strstr(rdsp_last_error(), "before fmt"));
  }
  remove(p1);
  remove(p2);
  remove(p3);
}

/* ---- designs ---------------------------------------------------------------------------- */
static void design_checks(void) {
  static const int sizes[] = {256, 512, 1024, 2048, 4096};
  float h_nat[256], hc[256];
  CHECK(rdsp_design_decimator(255, 10000.0, 96000.0, 1, h_nat, hc) != 0);
  CHECK(rdsp_design_decimator(256, 10000.0, 96000.0, 1, h_nat, hc) == 0);
  for (int k = 0; k < 256; k++) CHECK(hc[(k & 3) * 64 + (k >> 2)] == h_nat[k] && fabsf(h_nat[k] - h_nat[255 - k]) < 1e-9f);
  CHECK(rdsp_plan_radix(128) == 0 && rdsp_plan_radix(8192) == 0);
  for (size_t s = 0; s < sizeof sizes / sizeof sizes[0]; s++) {
    const int n = sizes[s], ntaps = n / 2 + 1;
    CHECK(rdsp_plan_radix(n) > 0);
    /* exactly sized buffers, so that one element too many is a sanitizer report */
    double *ci = (double *)malloc(sizeof(double) * (size_t)ntaps), *cq = (double *)malloc(sizeof(double) * (size_t)ntaps);
    float *mask = (float *)malloc(sizeof(float) * 2 * (size_t)n), *img = (float *)malloc(sizeof(float) * 2 * (size_t)n);
    unsigned char *seen = (unsigned char *)calloc((size_t)n, 1);
    CHECK(ci && cq && mask && img && seen);
    for (int win = 0; win <= 5; win++) {
      rdsp_calc_cplx_FIR_coeffs(ci, cq, ntaps, 300.0, 2700.0, 24000.0, win);
      for (int i = 0; i < ntaps; i++) CHECK(isfinite(ci[i]) && isfinite(cq[i]));
    }
    rdsp_calc_cplx_FIR_coeffs(ci, cq, ntaps, -2700.0, -300.0, 24000.0, 1);
    CHECK(rdsp_init_filter_mask(mask, ci, cq, n) == 0);
    rdsp_mask_device_image(mask, n, img);
    for (int i = 0; i < n; i++) {
      const int b = rdsp_bin_of_pos(n, i);
      CHECK(b >= 0 && b < n && !seen[b]);
      seen[b] = 1;
    }
    /* the image is the mask permuted and scaled */
    double sa = 0.0, sb = 0.0;
    for (int i = 0; i < 2 * n; i++) { sa += fabs((double)mask[i]); sb += fabs((double)img[i]); }
    CHECK(fabs(sa - sb * n) <= 1e-6 * sa); /* the image carries the 1/N of the inverse transform */
    if (n == 512) {
      float *fd = (float *)malloc(sizeof(float) * 2 * 4 * (size_t)n);
      CHECK(fd && rdsp_fd_decimator_image(h_nat, n, fd) == 0);
      double dc = 0.0; /* bin 0 of the four branches adds up to sum(h)/N */
      for (int r = 0; r < 4; r++) dc += fd[2 * ((size_t)r * (size_t)n)];
      double hs = 0.0;
      for (int k = 0; k < 256; k++) hs += h_nat[k];
      CHECK(fabs(dc * n - hs) < 1e-5);
      free(fd);
    }
    free(ci); free(cq); free(mask); free(img); free(seen);
  }
  CHECK(rdsp_init_filter_mask(h_nat, NULL, NULL, 100) != 0);
  CHECK(rdsp_fd_decimator_image(h_nat, 100, hc) != 0);
  float c5[5], c20[20];
  for (int kind = 0; kind < 4; kind++) {
    rdsp_biquad_design(kind, 500.0, 0.5, 24000.0, c5);
    for (int i = 0; i < 5; i++) CHECK(isfinite(c5[i]));
    int32_t t5[5]; /* the Teensy library's setters: 2.30 fixed point, feedback terms as the difference equation writes them */
    rdsp_teensy_biquad_design(kind, 500.0f, 0.5f, 44100.0f, t5);
    CHECK(abs(t5[4]) < (1 << 30) && t5[3] < 0 && t5[3] > -INT32_MAX);
  }
  rdsp_design_audio_iir(300.0, 2700.0, 24000.0, c20);
  for (int s = 0; s < 4; s++) {
    /* poles inside the unit circle for the stored (negated) feedback terms */
    const double a1 = -c20[5 * s + 3], a2 = -c20[5 * s + 4];
    CHECK(fabs(a2) < 1.0 && fabs(a1) < 1.0 + a2);
  }
  float rot[2], g1, g2, wmin, wmax;
  const uint32_t dphi = rdsp_nco_dphi(-12000.0, 96000.0);
  CHECK(dphi == 0xE0000000u);
  rdsp_nco_rot(dphi, 4, rot);
  CHECK(fabsf(rot[0] + 1.0f) < 1e-6f && fabsf(rot[1]) < 1e-6f);
  for (int s = -3; s <= 60; s++) CHECK(isfinite(rdsp_lms_mu(s)) && rdsp_lms_mu(s) > 0.0f);
  rdsp_sam_constants(24000.0, &g1, &g2, &wmin, &wmax);
  CHECK(g1 > 0.0f && g2 > 0.0f && wmin < 0.0f && wmax > 0.0f);
  /* the generator: a slice of a larger request is the same samples, with and without threads */
  enum { NCH = 3, NS = 700 };
  static int16_t a[NCH * NS * 2], b[2 * 300 * 2];
  rdsp_synth_config_t cfg = {96000.0, 12000.0, 0, 0.20, 0.30, 0.05};
  rdsp_synth_iq(a, 0, NCH, 0, NS, &cfg, 1);
  rdsp_synth_iq(b, 1, 2, 250, 300, &cfg, 4);
  for (int c = 0; c < 2; c++) CHECK(memcmp(b + (size_t)c * 600, a + ((size_t)(c + 1) * NS + 250) * 2, 600 * sizeof(int16_t)) == 0);
  cfg.cw = 1;
  rdsp_synth_iq(b, 5, 1, 1u << 20, 300, &cfg, 2);
}

/* ---- the analysers' tables and accessors (rdsp_q15_tables.c), exactly sized buffers ------------------- */
void rdsp_q15_twiddles(int n, uint32_t *out);
const uint16_t *rdsp_sqrt_guess_table(void);
void rdsp_arm_sin_table(float *tab513);
static void table_checks(void) {
  for (int id = 0; id <= RDSP_WINDOW_TUKEY; id++) {
    for (int n = 256; n <= 1024; n *= 4) {
      int16_t *w = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
      rdsp_window_q15_n(id, n, w);
      CHECK(id == RDSP_WINDOW_NONE ? w[n / 2] == 32767 : w[0] <= 3000);
      free(w);
    }
  }
  for (int n = 256; n <= 1024; n *= 4) {
    uint32_t *t = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(3 * n / 4));
    rdsp_q15_twiddles(n, t);
    CHECK((t[0] & 0xFFFFu) == 32767u && (t[0] >> 16) == 0u);
    free(t);
  }
  CHECK(rdsp_sqrt_guess_table()[0] == 55109 && rdsp_sqrt_guess_table()[32] == 0);
  CHECK(rdsp_sqrt_uint32_approx(0) == 0 && rdsp_sqrt_uint32_approx(1) == 1 && rdsp_sqrt_uint32_approx(0xFFFFFFFFu) >= 65535u);
  for (uint32_t x = 1; x; x <<= 1) CHECK(rdsp_sqrt_uint32_approx(x) > 0 && rdsp_sqrt_uint32_approx(x - 1 + (x == 1)) <= 65543u);
  float *tab = (float *)malloc(sizeof(float) * 513);
  rdsp_arm_sin_table(tab);
  CHECK(tab[0] == 0.0f && tab[128] == 1.0f && tab[384] == -1.0f && fabsf(tab[512]) < 1e-7f);
  free(tab);
  uint16_t *row = (uint16_t *)malloc(sizeof(uint16_t) * 256), *row2 = (uint16_t *)malloc(sizeof(uint16_t) * 512);
  for (int i = 0; i < 256; i++) row[i] = (uint16_t)(65535 - i);
  for (int i = 0; i < 512; i++) row2[i] = (uint16_t)i;
  /* every corner of the two read() forms on exactly sized rows: first == last, swapped, clamped, out of range */
  CHECK(rdsp_spectrum_read(row, 255) > 0.f && rdsp_spectrum_read(row, 256) == 0.f);
  CHECK(rdsp_spectrum_read_range(row, 255, 255) == rdsp_spectrum_read(row, 255));
  CHECK(rdsp_spectrum_read_range(row, 254, 9999) == rdsp_spectrum_read(row, 254));          /* clamp to 255, 255 not added */
  CHECK(rdsp_spectrum_read_range(row, 9999, 254) == rdsp_spectrum_read(row, 254));
  CHECK(rdsp_spectrum_read_range(row, 0, 255) > 0.f && rdsp_spectrum_read_range(row, 256, 300) == 0.f);
  CHECK(rdsp_fft1024_read(row2, 511) > 0.f && rdsp_fft1024_read(row2, 512) == 0.f);
  CHECK(rdsp_fft1024_read_range(row2, 511, 511) == rdsp_fft1024_read(row2, 511));
  CHECK(rdsp_fft1024_read_range(row2, 510, 9999) == (float)(510 + 511) * (float)(1.0 / 16384.0));
  CHECK(rdsp_fft1024_read_range(row2, 0, 511) > 0.f && rdsp_fft1024_read_range(row2, 512, 600) == 0.f);
  free(row);
  free(row2);
}

int main(int argc, char **argv) {
  if (argc != 2) {
    fprintf(stderr, "usage: host_sanitize <scratch directory>\n");
    return 64;
  }
  graph_checks(1);
  graph_checks(7);
  queue_ring_checks();
  io_checks(argv[1]);
  design_checks();
  table_checks();
  puts("host_sanitize OK");
  return 0;
}
