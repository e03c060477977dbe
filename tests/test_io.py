"""recorded-IQ reader, audio writer, streaming runner.

CPU part: the file layer against independent implementations (Python's `wave`
module and numpy's raw int16 reader) -- byte-exact, since this is integer data.
GPU part: files -> rdsp_stream_run_files -> files equals (bit for bit) what the
chain gives for the same samples resident in HBM, and follows the oracle within
+-1 LSB of the int16 audio (float parity is covered by test_gpu_parity.py)."""
import os
import struct
import wave

import numpy as np
import pytest

from cases import K1, K3


def _write_wav_py(path, iq, rate):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(rate)
        w.writeframes(np.ascontiguousarray(iq, dtype="<i2").tobytes())


def test_reader_raw_and_wav_byte_exact(rdsp, tmp_path):
    from radiodsp_sdr_rx_amd.io import IO_RAW, IO_WAV, IqReader
    rng = np.random.default_rng(11)
    iq = rng.integers(-32768, 32768, size=(5000, 2), dtype=np.int16)
    raw = tmp_path / "a.iq"
    iq.astype("<i2").tofile(raw)
    r = IqReader(raw)
    assert r.format == IO_RAW and r.frames == 5000 and r.sample_rate == 0.0
    got = np.concatenate([r.read(1234), r.read(1234), r.read(5000)])
    assert np.array_equal(got, iq) and len(r.read(10)) == 0
    wav = tmp_path / "a.wav"
    _write_wav_py(wav, iq, 96000)
    r = IqReader(wav)
    assert r.format == IO_WAV and r.frames == 5000 and r.sample_rate == 96000.0
    assert np.array_equal(r.read(6000), iq)


def test_reader_wav_with_extra_chunks_and_extensible_header(rdsp, tmp_path):
    """LIST chunk before fmt, odd-sized chunk padding, WAVE_FORMAT_EXTENSIBLE, unfinalised data size."""
    from radiodsp_sdr_rx_amd.io import IqReader
    iq = (np.arange(600, dtype=np.int16).reshape(300, 2) - 300)
    data = iq.astype("<i2").tobytes()
    fmt_ext = struct.pack("<HHIIHHHHIH14s", 0xFFFE, 2, 48000, 48000 * 4, 4, 16, 22, 16, 3, 1,
                          bytes.fromhex("000000001000800000aa00389b71"))
    body = b"WAVE" + b"LIST" + struct.pack("<I", 5) + b"abcde\0" + b"fmt " + struct.pack("<I", len(fmt_ext)) + fmt_ext
    body += b"data" + struct.pack("<I", 0xFFFFFFFF) + data          # recorder never patched the size
    p = tmp_path / "x.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", 0xFFFFFFFF) + body)
    r = IqReader(p)
    assert r.sample_rate == 48000.0 and r.frames == -1
    assert np.array_equal(r.read(1000), iq)


def test_reader_rejects_what_it_cannot_play(rdsp, tmp_path):
    from radiodsp_sdr_rx_amd import RdspError
    from radiodsp_sdr_rx_amd.io import IO_WAV, IqReader
    with pytest.raises(RdspError):
        IqReader(tmp_path / "missing.iq")
    mono = tmp_path / "mono.wav"
    with wave.open(str(mono), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(8000); w.writeframes(b"\0\0" * 10)
    with pytest.raises(RdspError):
        IqReader(mono)
    junk = tmp_path / "junk.wav"
    junk.write_bytes(b"not a wave file at all")
    with pytest.raises(RdspError):
        IqReader(junk, IO_WAV)


def test_writer_wav_is_readable_by_the_wave_module(rdsp, tmp_path):
    from radiodsp_sdr_rx_amd.io import IO_RAW, IO_WAV, AudioWriter
    rng = np.random.default_rng(12)
    lr = rng.integers(-32768, 32768, size=(3001, 2), dtype=np.int16)
    p = tmp_path / "o.wav"
    w = AudioWriter(p, IO_WAV, 24000.0)
    assert w.write(lr[:1000]) == 1000 and w.write(lr[1000:]) == 2001 and w.frames == 3001
    w.close()
    with wave.open(str(p), "rb") as f:
        assert (f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()) == (2, 2, 24000, 3001)
        assert np.array_equal(np.frombuffer(f.readframes(3001), "<i2").reshape(-1, 2), lr)
    assert os.path.getsize(p) == 44 + 3001 * 4
    q = tmp_path / "o.raw"
    w = AudioWriter(q, IO_RAW)
    w.write(lr); w.close()
    assert np.array_equal(np.fromfile(q, "<i2").reshape(-1, 2), lr)


# ---- GPU: the streaming runner ------------------------------------------------------------
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.mark.gpu
@pytest.mark.parametrize("name,cfg,pipelined", [("k1", K1, False), ("k3", K3, True)])
def test_files_through_the_runner_match_resident_processing_and_oracle(rdsp, oracle, torch_cuda, tmp_path, name, cfg, pipelined):
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.io import IO_RAW, IO_WAV, AudioWriter, IqReader, stream_files
    nch, per, calls = 3, 16, 5
    nblk = per * calls
    iq = synth_iq(nch, nblk * 128 + 300)       # 300 trailing samples: two blocks and a partial one, never a granule
    readers, writers = [], []
    for c in range(nch):
        if c % 2 == 0:
            p = tmp_path / f"in{c}.wav"
            _write_wav_py(p, iq[c], 96000)
        else:
            p = tmp_path / f"in{c}.iq"
            iq[c].astype("<i2").tofile(p)
        readers.append(IqReader(p))
        writers.append(AudioWriter(tmp_path / f"out{c}.wav", IO_WAV, 24000.0) if c % 2 == 0
                       else AudioWriter(tmp_path / f"out{c}.raw", IO_RAW))
    ch = Chain(nch, max_blocks_per_call=per, **cfg)
    ch.set_pipelined(pipelined)
    st = stream_files(ch, readers, writers, per)
    for w in writers:
        w.close()
    gran = 8 if cfg["fft_l"] <= 512 else cfg["fft_l"] // 64
    exp_blocks = (iq.shape[1] // 128) // gran * gran
    assert st["blocks"] == exp_blocks and st["samples_out"] == exp_blocks * 32
    got = []
    for c in range(nch):
        if c % 2 == 0:
            with wave.open(str(tmp_path / f"out{c}.wav"), "rb") as f:
                assert f.getframerate() == 24000
                got.append(np.frombuffer(f.readframes(f.getnframes()), "<i2").reshape(-1, 2))
        else:
            got.append(np.fromfile(tmp_path / f"out{c}.raw", "<i2").reshape(-1, 2))
    got = np.stack(got)
    assert got.shape == (nch, exp_blocks * 32, 2)
    # the same samples resident in HBM, one call per batch of `per` blocks + the tail batch
    ref_chain = Chain(nch, max_blocks_per_call=per, **cfg)
    parts, pos = [], 0
    while pos < exp_blocks:
        take = min(per, exp_blocks - pos)
        dev = torch.from_numpy(np.ascontiguousarray(iq[:, pos * 128:(pos + take) * 128])).cuda()
        parts.append(ref_chain.process(dev).cpu().numpy())
        pos += take
    assert np.array_equal(got, np.concatenate(parts, 1))
    # and the oracle on the int16 audio: the feed-forward chain directly (one LSB at rounding
    # boundaries); the chain with the NLMS stage is bit-identical to resident processing (above), which
    # the truth-anchored tests of test_gpu_parity.py hold against the oracle and the float64 model
    if name == "k1":
        for c in range(nch):
            r16 = oracle.OracleChain(**cfg).process(iq[c, :exp_blocks * 128])[0]
            assert np.abs(got[c].astype(np.int32) - r16.astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_memory_runner_and_callbacks(rdsp, torch_cuda):
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.io import stream_callbacks, stream_memory
    nch, nblk, per = 4, 64, 8
    iq = synth_iq(nch, nblk * 128)
    out, st = stream_memory(Chain(nch, max_blocks_per_call=per, **K1), iq, per)
    # the same samples resident in HBM in ONE call: the runner picks its own batches, and the default
    # decimator (one granule per frame) gives the same bits however a stream is cut into calls (CONV:231-245: fixed blocks)
    ref = Chain(nch, max_blocks_per_call=nblk, **K1).process(torch.from_numpy(iq).cuda()).cpu().numpy()
    assert st["blocks"] == nblk and np.array_equal(out, ref)
    # the throughput form (448-sample frames): the same batches give the same bits, one call differs by rounding
    fd = Chain(nch, max_blocks_per_call=per, fir_variant=2, **K1)
    out_fd, _ = stream_memory(fd, iq, per)
    rc = Chain(nch, max_blocks_per_call=per, fir_variant=2, **K1)
    ref_fd = np.concatenate([rc.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * per * 128:(k + 1) * per * 128])).cuda()).cpu().numpy()
                             for k in range(nblk // per)], axis=1)
    assert np.array_equal(out_fd, ref_fd) and np.abs(out_fd.astype(np.int32) - ref.astype(np.int32)).max() <= 1
    # page-locked arrays at both ends: the zero-copy path (DMA straight from / to the user's memory)
    pin = torch.from_numpy(iq).pin_memory()
    out_p, st = stream_memory(Chain(nch, max_blocks_per_call=per, **K1), pin, per)
    assert st["blocks"] == nblk and out_p.is_pinned() and np.array_equal(out_p.numpy(), ref)
    # generic callbacks: a source that dries up mid-batch (5 of 8 blocks: not a granule -> dropped)
    state = {"pos": 0}
    chunks = []

    def source(dst):
        n = min(dst.shape[1] // 128, (nblk - 3) - state["pos"])
        dst[:, :n * 128] = iq[:, state["pos"] * 128:(state["pos"] + n) * 128]
        state["pos"] += n
        return n

    st = stream_callbacks(Chain(nch, max_blocks_per_call=per, **K1), source, chunks.append, per)
    assert st["blocks"] == 56
    assert np.array_equal(np.concatenate(chunks, 1), ref[:, :56 * 32])


@pytest.mark.gpu
def test_runner_argument_errors(rdsp, torch_cuda):
    from radiodsp_sdr_rx_amd import RdspError
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.io import stream_memory
    iq = synth_iq(2, 16 * 128)
    with pytest.raises(RdspError):
        stream_memory(Chain(2, max_blocks_per_call=16, **K1), iq, 12)     # not a granule multiple


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8])
def test_memory_runner_random_shapes_match_resident_processing(rdsp, torch_cuda, seed):
    """the runner's three streams and two slots over random batch sizes, stream lengths (a ragged last batch,
    trailing blocks short of a granule), channel counts and chains, pipelined or not, pageable or page-locked
    arrays: bit-identical to the same batches processed resident in HBM"""
    torch = torch_cuda
    from cases import K3
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.io import stream_memory
    rng = np.random.default_rng(seed)
    cfg = [K1, K3, dict(fft_l=1024, demod="LSB", flo_hz=-2700.0, fhi_hz=-300.0, agc_mode="fast")][seed % 3]
    gran = max(8, cfg["fft_l"] // 64)
    nch = int(rng.integers(1, 40))
    per = gran * int(rng.integers(1, 5))
    nblk = int(rng.integers(per, 9 * per)) + int(rng.integers(0, gran))     # not a multiple of anything in general
    piped = bool(rng.integers(0, 2))
    iq = synth_iq(nch, nblk * 128)
    src = torch.from_numpy(iq).pin_memory() if seed % 2 else iq
    ch = Chain(nch, max_blocks_per_call=per, **cfg)
    ch.set_pipelined(piped)
    out, st = stream_memory(ch, src, per)
    out = out.numpy() if hasattr(out, "numpy") else out
    exp = nblk // gran * gran
    assert st["blocks"] == exp and st["samples_out"] == exp * 32
    out = out[:, :exp * 32]            # the array is sized for the whole input; trailing blocks short of a granule are not processed
    rc = Chain(nch, max_blocks_per_call=per, **cfg)
    rc.set_pipelined(piped)
    parts, pos = [], 0
    while pos < exp:
        take = min(per, exp - pos)
        parts.append(rc.process(torch.from_numpy(np.ascontiguousarray(iq[:, pos * 128:(pos + take) * 128])).cuda()))
        pos += take
    rc.flush()
    torch.cuda.synchronize()
    assert np.array_equal(out, np.concatenate([p.cpu().numpy() for p in parts], 1))


def test_iq_slip_estimate_of_a_recording(rdsp):
    """rdsp_estimate_iq_slip (host C, no GPU): a stream with a dominant one-sided line loses its image
    rejection when one rail is a sample late; the estimate names the correction for rdsp_pre_setIQslip."""
    from radiodsp_sdr_rx_amd.chain import estimate_iq_slip, synth_iq
    from radiodsp_sdr_rx_amd import RdspError
    iq = synth_iq(2, 6000)
    for c in range(2):
        slip, rej = estimate_iq_slip(iq[c])
        assert slip == 0 and rej[0] > 30.0 and max(rej[1:]) < 15.0
        q_late = iq[c].copy(); q_late[1:, 1] = iq[c, :-1, 1]
        i_late = iq[c].copy(); i_late[1:, 0] = iq[c, :-1, 0]
        assert estimate_iq_slip(q_late)[0] == 1 and estimate_iq_slip(i_late)[0] == -1
    # no dominant one-sided line: white noise, a real-valued signal (Q = 0), a double-side-band tone.  All three
    # hypotheses sit near 0 dB there; no correction may be recommended (a spurious +-1 would wreck a healthy recording)
    rng = np.random.default_rng(9)
    t = np.arange(6000)
    noise = rng.integers(-8000, 8000, size=(6000, 2)).astype(np.int16)
    real = np.stack([(9000 * np.sin(2 * np.pi * 0.07 * t) + rng.normal(0, 300, 6000)), np.zeros(6000)], 1).astype(np.int16)
    dsb = np.stack([9000 * np.cos(2 * np.pi * 0.11 * t), 9000 * np.cos(2 * np.pi * 0.11 * t + 0.3)], 1).astype(np.int16)
    for name, sig in (("noise", noise), ("real", real), ("dsb", dsb)):
        slip, rej = estimate_iq_slip(sig)
        assert slip == 0, (name, slip, rej)
    with pytest.raises(RdspError):
        estimate_iq_slip(iq[0, :100])
