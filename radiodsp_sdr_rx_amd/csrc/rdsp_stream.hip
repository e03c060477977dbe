/*
 * rdsp_stream.hip -- streaming runner: host source -> HBM ->
 * receive chain -> HBM -> host sink, what loop() + the I2S DMA do in the sketch
 * (RadioDSP_SDR_RX.ino:195-198; queues RDSP_convolutional.h:231-244,344-349),
 * for recorded IQ.  Three HIP streams (upload, compute, download) over two
 * slots of pinned host memory, so that reading the next batch, the PCIe copies
 * and the kernels of consecutive batches overlap; the host only ever waits for
 * the download of the batch before the one it just queued.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "rdsp_host.h"

#define HIP_TRYS(expr)                                                             \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      rdsp_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      rc = RDSP_ERR_HIP;                                                           \
      goto done;                                                                   \
    }                                                                              \
  } while (0)

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

extern "C" int rdsp_stream_run(rdsp_chain_t *c, rdsp_source_fn source, void *source_user, rdsp_sink_fn sink,
                               void *sink_user, int blocks_per_call, int64_t max_blocks,
                               rdsp_stream_stats_t *stats) {
  if (!c || !source || !sink || blocks_per_call <= 0) {
    rdsp_set_error("rdsp_stream_run: bad argument");
    return RDSP_ERR_INVALID;
  }
  const int gran = rdsp_chain_call_unit_blocks(c);
  if (blocks_per_call % gran != 0) {
    rdsp_set_error("blocks_per_call %d is not a multiple of the call unit %d", blocks_per_call, gran);
    return RDSP_ERR_NOT_READY;
  }
  /* the runner's buffers, streams and events live on the chain's device, whatever the calling thread had
   * selected (a host that drives several GPUs from one process) */
  if (hipSetDevice(rdsp_chain_device(c)) != hipSuccess) {
    rdsp_set_error("hipSetDevice(%d) failed", rdsp_chain_device(c));
    return RDSP_ERR_HIP;
  }
  const int nch = rdsp_chain_channels(c);
  const int decim = rdsp_chain_decim(c);
  const size_t in_stride = (size_t)blocks_per_call * RDSP_BLOCK_SAMPLES; /* IQ pairs per channel row */
  const size_t out_stride = in_stride / (size_t)decim;
  const size_t in_bytes = in_stride * 4 * (size_t)nch, out_bytes = out_stride * 4 * (size_t)nch;

  int rc = RDSP_OK;
  int16_t *hin[2] = {nullptr, nullptr}, *hout[2] = {nullptr, nullptr}, *din[2] = {nullptr, nullptr},
          *dout[2] = {nullptr, nullptr};
  hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
  int out_pairs[2] = {0, 0};
  int64_t done_blocks = 0;
  double t_read = 0.0, t_write = 0.0;
  const double t_begin = now_s();
  bool ended = false;
  int64_t it = 0;

  for (int i = 0; i < 2; i++) {
    HIP_TRYS(hipHostMalloc((void **)&hin[i], in_bytes, hipHostMallocDefault));
    HIP_TRYS(hipHostMalloc((void **)&hout[i], out_bytes, hipHostMallocDefault));
    HIP_TRYS(hipMalloc((void **)&din[i], in_bytes));
    HIP_TRYS(hipMalloc((void **)&dout[i], out_bytes));
    HIP_TRYS(hipEventCreateWithFlags(&ev_up[i], hipEventDisableTiming));
    HIP_TRYS(hipEventCreateWithFlags(&ev_comp[i], hipEventDisableTiming));
    HIP_TRYS(hipEventCreateWithFlags(&ev_down[i], hipEventDisableTiming));
  }
  HIP_TRYS(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
  HIP_TRYS(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
  HIP_TRYS(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));

  for (; !ended; it++) {
    const int slot = (int)(it & 1);
    int want = blocks_per_call;
    if (max_blocks > 0 && max_blocks - done_blocks < (int64_t)want) want = (int)(max_blocks - done_blocks);
    want -= want % gran;
    if (want <= 0) break;
    if (it >= 2) HIP_TRYS(hipEventSynchronize(ev_up[slot])); /* the pinned input slot is free again */
    double t0 = now_s();
    int got = source(source_user, hin[slot], in_stride, want); /* Q_in_L/R.readBuffer(), CONV:236-244 */
    t_read += now_s() - t0;
    if (got < 0) {
      rdsp_set_error("stream source failed (%d)", got);
      rc = RDSP_ERR_INVALID;
      goto done;
    }
    if (got < want) ended = true; /* the sketch would keep waiting for a full granule (CONV:231): stop */
    got -= got % gran;
    if (got > 0) {
      /* upload: the compute of two batches ago has finished reading din[slot] */
      if (it >= 2) HIP_TRYS(hipStreamWaitEvent(s_up, ev_comp[slot], 0));
      if (got == blocks_per_call) {
        HIP_TRYS(hipMemcpyAsync(din[slot], hin[slot], in_bytes, hipMemcpyHostToDevice, s_up));
      } else {
        HIP_TRYS(hipMemcpy2DAsync(din[slot], in_stride * 4, hin[slot], in_stride * 4, (size_t)got * RDSP_BLOCK_SAMPLES * 4,
                                  (size_t)nch, hipMemcpyHostToDevice, s_up));
      }
      HIP_TRYS(hipEventRecord(ev_up[slot], s_up));
      /* compute: needs the upload, and dout[slot] drained by the download of two batches ago */
      HIP_TRYS(hipStreamWaitEvent(s_comp, ev_up[slot], 0));
      if (it >= 2) HIP_TRYS(hipStreamWaitEvent(s_comp, ev_down[slot], 0));
      rc = rdsp_chain_process(c, din[slot], in_stride, got, dout[slot], out_stride, nullptr, s_comp);
      if (rc != RDSP_OK) goto done;
      HIP_TRYS(hipEventRecord(ev_comp[slot], s_comp));
      /* download: after the whole chain of this batch (pipelined mode: its tail stage) */
      HIP_TRYS(hipStreamWaitEvent(s_down, ev_comp[slot], 0));
      rc = rdsp_chain_flush(c, s_down);
      if (rc != RDSP_OK) goto done;
      out_pairs[slot] = got * RDSP_BLOCK_SAMPLES / decim;
      if (got == blocks_per_call) {
        HIP_TRYS(hipMemcpyAsync(hout[slot], dout[slot], out_bytes, hipMemcpyDeviceToHost, s_down));
      } else {
        HIP_TRYS(hipMemcpy2DAsync(hout[slot], out_stride * 4, dout[slot], out_stride * 4, (size_t)out_pairs[slot] * 4,
                                  (size_t)nch, hipMemcpyDeviceToHost, s_down));
      }
      HIP_TRYS(hipEventRecord(ev_down[slot], s_down));
      done_blocks += got;
    } else {
      out_pairs[slot] = 0;
    }
    /* hand the previous batch to the sink while this one is in flight */
    if (it >= 1 && out_pairs[slot ^ 1] > 0) {
      HIP_TRYS(hipEventSynchronize(ev_down[slot ^ 1]));
      t0 = now_s();
      const int w = sink(sink_user, hout[slot ^ 1], out_stride, out_pairs[slot ^ 1]); /* Q_out_L/R.playBuffer(), CONV:344-349 */
      t_write += now_s() - t0;
      out_pairs[slot ^ 1] = 0;
      if (w < 0) {
        rdsp_set_error("stream sink failed (%d)", w);
        rc = RDSP_ERR_INVALID;
        goto done;
      }
    }
  }
  { /* the last batch */
    const int last = (int)((it - 1) & 1);
    if (it >= 1 && out_pairs[last] > 0) {
      HIP_TRYS(hipEventSynchronize(ev_down[last]));
      const double t0 = now_s();
      const int w = sink(sink_user, hout[last], out_stride, out_pairs[last]);
      t_write += now_s() - t0;
      if (w < 0) {
        rdsp_set_error("stream sink failed (%d)", w);
        rc = RDSP_ERR_INVALID;
      }
    }
  }

done:
  if (s_up) (void)hipStreamSynchronize(s_up);
  if (s_comp) (void)hipStreamSynchronize(s_comp);
  if (s_down) (void)hipStreamSynchronize(s_down);
  if (stats) {
    stats->blocks = done_blocks;
    stats->samples_in = done_blocks * RDSP_BLOCK_SAMPLES;
    stats->samples_out = done_blocks * RDSP_BLOCK_SAMPLES / decim;
    stats->seconds = now_s() - t_begin;
    stats->read_seconds = t_read;
    stats->write_seconds = t_write;
  }
  for (int i = 0; i < 2; i++) {
    if (hin[i]) (void)hipHostFree(hin[i]);
    if (hout[i]) (void)hipHostFree(hout[i]);
    if (din[i]) (void)hipFree(din[i]);
    if (dout[i]) (void)hipFree(dout[i]);
    if (ev_up[i]) (void)hipEventDestroy(ev_up[i]);
    if (ev_comp[i]) (void)hipEventDestroy(ev_comp[i]);
    if (ev_down[i]) (void)hipEventDestroy(ev_down[i]);
  }
  if (s_up) (void)hipStreamDestroy(s_up);
  if (s_comp) (void)hipStreamDestroy(s_comp);
  if (s_down) (void)hipStreamDestroy(s_down);
  return rc;
}

/* ---- files: one reader and one writer per channel ------------------------------------ */
struct FileEnds {
  rdsp_iq_reader_t *const *readers;
  rdsp_audio_writer_t *const *writers;
  int nch;
};

static int file_source(void *user, int16_t *dst, size_t stride_pairs, int n_blocks) {
  FileEnds *fe = (FileEnds *)user;
  const size_t want = (size_t)n_blocks * RDSP_BLOCK_SAMPLES;
  size_t least = want;
#pragma omp parallel for schedule(dynamic, 1) reduction(min : least)
  for (int ch = 0; ch < fe->nch; ch++) {
    const size_t got = rdsp_iq_reader_read(fe->readers[ch], dst + (size_t)ch * stride_pairs * 2, want);
    if (got < least) least = got;
  }
  return (int)(least / RDSP_BLOCK_SAMPLES); /* the shortest recording ends the run */
}

static int file_sink(void *user, const int16_t *src, size_t stride_pairs, int n_pairs) {
  FileEnds *fe = (FileEnds *)user;
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : bad)
  for (int ch = 0; ch < fe->nch; ch++)
    if (rdsp_audio_writer_write(fe->writers[ch], src + (size_t)ch * stride_pairs * 2, (size_t)n_pairs) != (size_t)n_pairs) bad++;
  return bad ? -1 : n_pairs;
}

extern "C" int rdsp_stream_run_files(rdsp_chain_t *c, rdsp_iq_reader_t *const *readers,
                                     rdsp_audio_writer_t *const *writers, int blocks_per_call,
                                     int64_t max_blocks, rdsp_stream_stats_t *stats) {
  if (!c || !readers || !writers) return RDSP_ERR_INVALID;
  FileEnds fe = {readers, writers, rdsp_chain_channels(c)};
  for (int i = 0; i < fe.nch; i++)
    if (!readers[i] || !writers[i]) {
      rdsp_set_error("channel %d has no reader/writer", i);
      return RDSP_ERR_INVALID;
    }
  return rdsp_stream_run(c, file_source, &fe, file_sink, &fe, blocks_per_call, max_blocks, stats);
}

/* ---- memory: host arrays [n_channels][stride] at both ends ------------------------------ */
struct MemEnds {
  const int16_t *in;
  size_t in_stride;
  int64_t blocks_left, in_pos;
  int16_t *out;
  size_t out_stride;
  int64_t out_pos;
  int nch;
};

static int mem_source(void *user, int16_t *dst, size_t stride_pairs, int n_blocks) {
  MemEnds *m = (MemEnds *)user;
  const int take = (int)((int64_t)n_blocks < m->blocks_left ? (int64_t)n_blocks : m->blocks_left);
  const size_t pairs = (size_t)take * RDSP_BLOCK_SAMPLES;
#pragma omp parallel for schedule(static)
  for (int ch = 0; ch < m->nch; ch++)
    memcpy(dst + (size_t)ch * stride_pairs * 2, m->in + ((size_t)ch * m->in_stride + (size_t)m->in_pos) * 2, pairs * 4);
  m->in_pos += (int64_t)pairs;
  m->blocks_left -= take;
  return take;
}

static int mem_sink(void *user, const int16_t *src, size_t stride_pairs, int n_pairs) {
  MemEnds *m = (MemEnds *)user;
#pragma omp parallel for schedule(static)
  for (int ch = 0; ch < m->nch; ch++)
    memcpy(m->out + ((size_t)ch * m->out_stride + (size_t)m->out_pos) * 2, src + (size_t)ch * stride_pairs * 2, (size_t)n_pairs * 4);
  m->out_pos += n_pairs;
  return n_pairs;
}

/* both arrays page-locked (hipHostMalloc / hipHostRegister / torch pin_memory): the DMA engines
 * read and write them directly, no staging slots and no host copies */
static bool is_pinned_host(const void *p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeHost;
}

static int stream_pinned(rdsp_chain_t *c, const int16_t *host_iq, size_t in_stride, int64_t n_blocks, int16_t *host_out,
                         size_t out_stride, int blocks_per_call, rdsp_stream_stats_t *stats) {
  const int gran = rdsp_chain_call_unit_blocks(c);
  if (blocks_per_call <= 0 || blocks_per_call % gran != 0) {
    rdsp_set_error("blocks_per_call %d is not a multiple of the call unit %d", blocks_per_call, gran);
    return RDSP_ERR_NOT_READY;
  }
  if (hipSetDevice(rdsp_chain_device(c)) != hipSuccess) {
    rdsp_set_error("hipSetDevice(%d) failed", rdsp_chain_device(c));
    return RDSP_ERR_HIP;
  }
  const int nch = rdsp_chain_channels(c), decim = rdsp_chain_decim(c);
  const size_t d_in = (size_t)blocks_per_call * RDSP_BLOCK_SAMPLES, d_out = d_in / (size_t)decim;
  int rc = RDSP_OK;
  int16_t *din[2] = {nullptr, nullptr}, *dout[2] = {nullptr, nullptr};
  hipStream_t s_up = nullptr, s_comp = nullptr, s_down = nullptr;
  hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr}, ev_down[2] = {nullptr, nullptr};
  int64_t done = 0, it = 0;
  const double t_begin = now_s();
  for (int i = 0; i < 2; i++) {
    HIP_TRYS(hipMalloc((void **)&din[i], d_in * 4 * (size_t)nch));
    HIP_TRYS(hipMalloc((void **)&dout[i], d_out * 4 * (size_t)nch));
    HIP_TRYS(hipEventCreateWithFlags(&ev_up[i], hipEventDisableTiming));
    HIP_TRYS(hipEventCreateWithFlags(&ev_comp[i], hipEventDisableTiming));
    HIP_TRYS(hipEventCreateWithFlags(&ev_down[i], hipEventDisableTiming));
  }
  HIP_TRYS(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
  HIP_TRYS(hipStreamCreateWithFlags(&s_comp, hipStreamNonBlocking));
  HIP_TRYS(hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking));
  for (; done < n_blocks; it++) {
    const int slot = (int)(it & 1);
    int take = (int)((n_blocks - done) < (int64_t)blocks_per_call ? (n_blocks - done) : (int64_t)blocks_per_call);
    take -= take % gran;
    if (take <= 0) break;
    const size_t pin = (size_t)take * RDSP_BLOCK_SAMPLES, pout = pin / (size_t)decim;
    if (it >= 2) HIP_TRYS(hipStreamWaitEvent(s_up, ev_comp[slot], 0)); /* din[slot] consumed */
    HIP_TRYS(hipMemcpy2DAsync(din[slot], d_in * 4, host_iq + (size_t)done * RDSP_BLOCK_SAMPLES * 2, in_stride * 4, pin * 4,
                              (size_t)nch, hipMemcpyHostToDevice, s_up));
    HIP_TRYS(hipEventRecord(ev_up[slot], s_up));
    HIP_TRYS(hipStreamWaitEvent(s_comp, ev_up[slot], 0));
    if (it >= 2) HIP_TRYS(hipStreamWaitEvent(s_comp, ev_down[slot], 0)); /* dout[slot] drained */
    rc = rdsp_chain_process(c, din[slot], d_in, take, dout[slot], d_out, nullptr, s_comp);
    if (rc != RDSP_OK) goto done;
    HIP_TRYS(hipEventRecord(ev_comp[slot], s_comp));
    HIP_TRYS(hipStreamWaitEvent(s_down, ev_comp[slot], 0));
    rc = rdsp_chain_flush(c, s_down);
    if (rc != RDSP_OK) goto done;
    HIP_TRYS(hipMemcpy2DAsync(host_out + (size_t)done * RDSP_BLOCK_SAMPLES / (size_t)decim * 2, out_stride * 4, dout[slot],
                              d_out * 4, pout * 4, (size_t)nch, hipMemcpyDeviceToHost, s_down));
    HIP_TRYS(hipEventRecord(ev_down[slot], s_down));
    done += take;
  }
done:
  if (s_up) (void)hipStreamSynchronize(s_up);
  if (s_comp) (void)hipStreamSynchronize(s_comp);
  if (s_down) (void)hipStreamSynchronize(s_down);
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->blocks = done;
    stats->samples_in = done * RDSP_BLOCK_SAMPLES;
    stats->samples_out = done * RDSP_BLOCK_SAMPLES / decim;
    stats->seconds = now_s() - t_begin;
  }
  for (int i = 0; i < 2; i++) {
    if (din[i]) (void)hipFree(din[i]);
    if (dout[i]) (void)hipFree(dout[i]);
    if (ev_up[i]) (void)hipEventDestroy(ev_up[i]);
    if (ev_comp[i]) (void)hipEventDestroy(ev_comp[i]);
    if (ev_down[i]) (void)hipEventDestroy(ev_down[i]);
  }
  if (s_up) (void)hipStreamDestroy(s_up);
  if (s_comp) (void)hipStreamDestroy(s_comp);
  if (s_down) (void)hipStreamDestroy(s_down);
  return rc;
}

extern "C" int rdsp_stream_run_memory(rdsp_chain_t *c, const int16_t *host_iq, size_t in_stride_pairs,
                                      int64_t n_blocks, int16_t *host_out, size_t out_stride_pairs,
                                      int blocks_per_call, rdsp_stream_stats_t *stats) {
  if (!c || !host_iq || !host_out || n_blocks <= 0) return RDSP_ERR_INVALID;
  const int decim = rdsp_chain_decim(c);
  if (in_stride_pairs < (size_t)n_blocks * RDSP_BLOCK_SAMPLES ||
      out_stride_pairs < (size_t)n_blocks * RDSP_BLOCK_SAMPLES / (size_t)decim) {
    rdsp_set_error("rdsp_stream_run_memory: strides too small");
    return RDSP_ERR_INVALID;
  }
  if (is_pinned_host(host_iq) && is_pinned_host(host_out))
    return stream_pinned(c, host_iq, in_stride_pairs, n_blocks, host_out, out_stride_pairs, blocks_per_call, stats);
  MemEnds m = {host_iq, in_stride_pairs, n_blocks, 0, host_out, out_stride_pairs, 0, rdsp_chain_channels(c)};
  return rdsp_stream_run(c, mem_source, &m, mem_sink, &m, blocks_per_call, n_blocks, stats);
}
