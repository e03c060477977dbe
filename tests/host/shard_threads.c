/*
 * shard_threads.c -- the host shape SURVEY 8(e) describes, in plain C over the boundary: independent
 * receiver channels sharded by contiguous ranges, one host thread + one chain + one stream per shard,
 * no exchange between shards.  Shard k runs on device k % (number of devices): on an 8-GPU node that is
 * one bucket per GPU; on a one-GPU box the shards share the device, which is what this test can check
 * here -- the library has no cross-chain state, so T threads driving T chains concurrently must give,
 * bit for bit, what one chain over all channels gives from one thread (channel partition invariance),
 * pipelined mode on, BASELINE config K3 (spectral NR + LMS notch + AGC).
 *
 *   shard_threads <shards> <channels per shard> <calls> <blocks per call>
 *
 * Built by tests/test_boundary_c.py: gcc -std=c11 -pthread -D__HIP_PLATFORM_AMD__ -I include
 *   -I /opt/rocm/include shard_threads.c -L radiodsp_sdr_rx_amd -lrdsp_hip -L /opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rdsp.h"

#define CHECK(x)                                                                      \
  do {                                                                                \
    if (!(x)) {                                                                       \
      fprintf(stderr, "%s:%d: %s failed (%s)\n", __FILE__, __LINE__, #x, rdsp_last_error()); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

static rdsp_chain_config_t k3_config(void) {
  rdsp_chain_config_t c;
  memset(&c, 0, sizeof(c));
  c.fs_in = 96000.0; c.decim = 4; c.fir_taps = 256; c.fir_cut_hz = 10000.0; c.nco_hz = 12000.0;
  c.fft_l = 512; c.window = 1; c.flo_hz = 300.0; c.fhi_hz = 2700.0; c.filter_on = 1;
  c.demod = RDSP_DEMOD_USB; c.spectral_nr = 1; c.spectral_level = 2.0f;
  c.als_mode = RDSP_ALS_NOTCH; c.als_strength = 20; c.agc_mode = RDSP_AGC_MEDIUM;
  c.input_gain = 1.0f; c.output_gain = 0.5f; c.iq_balance = 1.0f;
  return c;
}

typedef struct {
  int ch0, n_ch, calls, blocks, device;
  int16_t *audio; /* host, [n_ch][calls * blocks * 32][2] */
} shard_t;

/* one receiver bucket: its own chain, stream and buffers; channels ch0 .. ch0 + n_ch - 1 */
static void *run_shard(void *arg) {
  shard_t *s = (shard_t *)arg;
  const size_t in_pairs = (size_t)s->blocks * RDSP_BLOCK_SAMPLES, out_pairs = in_pairs / 4;
  const size_t total_out = (size_t)s->calls * out_pairs;
  CHECK(hipSetDevice(s->device) == hipSuccess);
  hipStream_t stream;
  CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess);
  rdsp_chain_config_t cfg = k3_config();
  rdsp_chain_t *chain = NULL;
  CHECK(rdsp_chain_create(&cfg, s->n_ch, s->device, s->blocks, &chain) == RDSP_OK);
  CHECK(rdsp_chain_set_pipelined(chain, 1) == RDSP_OK);
  int16_t *h_iq = NULL, *d_iq = NULL, *d_out = NULL;
  CHECK(hipHostMalloc((void **)&h_iq, (size_t)s->n_ch * in_pairs * 4, hipHostMallocDefault) == hipSuccess);
  CHECK(hipMalloc((void **)&d_iq, (size_t)s->n_ch * in_pairs * 4 * 2) == hipSuccess); /* two call slots */
  CHECK(hipMalloc((void **)&d_out, (size_t)s->n_ch * total_out * 4) == hipSuccess);
  rdsp_synth_config_t sc = {96000.0, 12000.0, 0, 0.20, 0.30, 0.05};
  for (int k = 0; k < s->calls; k++) {
    int16_t *slot = d_iq + (size_t)(k & 1) * s->n_ch * in_pairs * 2;
    CHECK(hipStreamSynchronize(stream) == hipSuccess); /* h_iq is free again (a host that cared would double-buffer it too) */
    rdsp_synth_iq(h_iq, s->ch0, s->n_ch, (uint64_t)k * in_pairs, (int)in_pairs, &sc, 1);
    CHECK(hipMemcpyAsync(slot, h_iq, (size_t)s->n_ch * in_pairs * 4, hipMemcpyHostToDevice, stream) == hipSuccess);
    CHECK(rdsp_chain_process(chain, slot, in_pairs, s->blocks, d_out + (size_t)k * out_pairs * 2, total_out, NULL,
                             stream) == RDSP_OK);
  }
  CHECK(rdsp_chain_flush(chain, stream) == RDSP_OK);
  CHECK(hipMemcpyAsync(s->audio, d_out, (size_t)s->n_ch * total_out * 4, hipMemcpyDeviceToHost, stream) == hipSuccess);
  CHECK(hipStreamSynchronize(stream) == hipSuccess);
  rdsp_chain_destroy(chain);
  (void)hipFree(d_out);
  (void)hipFree(d_iq);
  (void)hipHostFree(h_iq);
  (void)hipStreamDestroy(stream);
  return NULL;
}

int main(int argc, char **argv) {
  if (argc != 5) {
    fprintf(stderr, "usage: %s shards channels_per_shard calls blocks_per_call\n", argv[0]);
    return 64;
  }
  const int shards = atoi(argv[1]), per = atoi(argv[2]), calls = atoi(argv[3]), blocks = atoi(argv[4]);
  if (shards < 1 || shards > 64 || per < 1 || calls < 1 || blocks < 8 || blocks % 8) return 64;
  const int ndev = rdsp_device_count();
  if (ndev <= 0) {
    fprintf(stderr, "no HIP device: %s\n", rdsp_last_error());
    return 69;
  }
  const size_t out_per_ch = (size_t)calls * blocks * 32 * 2; /* int16 per channel */
  int16_t *sharded = (int16_t *)malloc((size_t)shards * per * out_per_ch * sizeof(int16_t));
  int16_t *whole = (int16_t *)malloc((size_t)shards * per * out_per_ch * sizeof(int16_t));
  CHECK(sharded && whole);
  shard_t sh[64];
  pthread_t th[64];
  for (int k = 0; k < shards; k++) {
    sh[k] = (shard_t){k * per, per, calls, blocks, k % ndev, sharded + (size_t)k * per * out_per_ch};
    CHECK(pthread_create(&th[k], NULL, run_shard, &sh[k]) == 0);
  }
  for (int k = 0; k < shards; k++) CHECK(pthread_join(th[k], NULL) == 0);
  /* the same channels as one bucket, from this thread */
  shard_t all = {0, shards * per, calls, blocks, 0, whole};
  run_shard(&all);
  size_t diff = 0;
  for (size_t i = 0; i < (size_t)shards * per * out_per_ch; i++) diff += sharded[i] != whole[i];
  long nonzero = 0;
  for (size_t i = 0; i < (size_t)shards * per * out_per_ch; i++) nonzero += whole[i] != 0;
  printf("shard_threads: %d shards x %d channels on %d device(s), %d calls x %d blocks: %zu of %zu int16 differ, %ld non-zero\n",
         shards, per, ndev, calls, blocks, diff, (size_t)shards * per * out_per_ch, nonzero);
  free(sharded);
  free(whole);
  return (diff == 0 && nonzero > 0) ? 0 : 1;
}
