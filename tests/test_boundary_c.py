"""The drop-in boundary exercised from C: tests/host/rdsp_binding.h is
the reference-side binding, tests/host/binding_check.c is setup()/loop() of the sketch over it.
CPU: it compiles and links against librdsp_hip.so with gcc.  GPU: it runs BASELINE config K1 and
its audio is bit-identical to the ctypes path and within 1 LSB of the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tests", "host")
CFG = dict(fft_l=256, demod="LSB", flo_hz=300.0, fhi_hz=4000.0, agc_mode="medium", input_gain=1.0, output_gain=0.5,
           iq_balance=1.02, nco_hz=8390.0)   # what setup() of binding_check.c leaves the engine in: LSBmode's TuningOffset
# (the AudioSDR engine's own answer, tests/golden/firmware_kat.npz) handed to the mixer by the binding


def build(tmp_path, name="binding_check", defines=(), out=None):
    exe = str(tmp_path / (out or name))
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-pthread", "-D__HIP_PLATFORM_AMD__"] + list(defines) +
                          ["-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
                           os.path.join(HOST, name + ".c"), "-o", exe,
                           "-L", os.path.join(ROOT, "radiodsp_sdr_rx_amd"), "-lrdsp_hip",
                           "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "radiodsp_sdr_rx_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_reference_side_binding_compiles_and_links_from_c(rdsp, tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 64 and "usage" in r.stderr      # argument check only: no compute without a GPU


@pytest.mark.gpu
def test_k1_through_the_c_binding_matches_ctypes_path_and_oracle(rdsp, oracle, tmp_path):
    import torch
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    nblk = 64
    iq = synth_iq(1, nblk * 128)
    fin, fout = tmp_path / "iq.raw", tmp_path / "audio.raw"
    iq[0].tofile(fin)
    exe = build(tmp_path)
    r = subprocess.run([exe, str(fin), str(fout), str(nblk)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "TuningOffset 8390" in r.stdout                 # LSBmode (INO:139): the engine's low IF, 6890 + 1500 Hz
    got = np.fromfile(fout, dtype=np.int16).reshape(-1, 2)
    assert got.shape == (nblk * 32, 2)
    ch = Chain(1, max_blocks_per_call=16, **CFG)
    ref = np.concatenate([ch.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * 2048:(k + 1) * 2048])).cuda()).cpu().numpy()[0]
                          for k in range(nblk // 16)])
    assert np.array_equal(got, ref)
    r16, _ = oracle.OracleChain(**CFG).process(iq[0])
    assert np.abs(got.astype(np.int32) - r16.astype(np.int32)).max() <= 1


@pytest.mark.gpu
def test_panadapter_side_through_the_c_binding_matches_the_oracle(rdsp, oracle, tmp_path):
    """The same C program with the panadapter side of the graph switched on (INO:57-59,75-78,144-145,155-156):
    biquad1 / biquad2 (high-pass 500 Hz, q 0.5) on the I and Q rails, AudioAnalyzeFFT256IQ with
    AudioWindowHanning256 handed over by pointer and averageTogether(30); every spectrum FFT.available() announces,
    FFT.read(80) and FFT.read(75, 85) come back through a file: bit-exact against the oracle's restatements."""
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from test_audio_nodes import TeensyBiquadOracle, _bind
    from test_spectrum import _olib, oracle_spectra
    lib = _bind(_olib(oracle))
    nblk = 128
    iq = synth_iq(1, nblk * 128)
    fin, fout, fspec = tmp_path / "iq.raw", tmp_path / "audio.raw", tmp_path / "spec.raw"
    iq[0].tofile(fin)
    exe = build(tmp_path)
    r = subprocess.run([exe, str(fin), str(fout), str(nblk), str(fspec)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    rec = np.fromfile(fspec, dtype=np.uint8).reshape(-1, 512 + 8)
    spectra = rec[:, :512].copy().view(np.uint16)
    reads = rec[:, 512:].copy().view(np.float32)
    filt = np.zeros((nblk * 128, 2), np.int16)
    for side in (0, 1):
        o = TeensyBiquadOracle(lib)                                   # the Teensy library's fixed-point AudioFilterBiquad
        o.set(0, "highpass", 500.0, 0.5)
        filt[:, side] = o.update(np.ascontiguousarray(iq[0, :, side]))
    want = np.stack(oracle_spectra(lib, filt, 30, 1))
    # the C program looks at the flag once per 16-block call: it reports the latest average of each call that completed one
    assert len(spectra) >= 3
    idx = [i for i in range(len(want)) if any(np.array_equal(want[i], s) for s in spectra)]
    assert len(idx) == len(spectra), (len(idx), len(spectra))
    for s, rd in zip(spectra, reads):
        assert rd[0] == np.float32(s[80]) * np.float32(1.0 / 16384.0)
        assert rd[1] == np.float32(int(s[75:85].sum())) * np.float32(1.0 / 16384.0)       # FFTIQ.h:75-86: bin 85 is not added


def test_literal_sketch_binding_compiles(rdsp, tmp_path):
    exe = build(tmp_path, defines=["-DRDSP_BIND_LITERAL"], out="binding_literal")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 64


@pytest.mark.gpu
@pytest.mark.parametrize("nr", [0, 15])
def test_the_sketch_over_the_c_binding_against_the_sketchs_own_binary(rdsp, tmp_path, nr):
    """setup() / loop() of the sketch over the reference-side binding, in C, with the CONV stage configured as the
    firmware image has it (-DRDSP_BIND_LITERAL: 44.1 kHz, FFT_L 256, 129 taps, 300 ... 4000 Hz, L / R through) and fed
    the IQ blocks the image's doConvolutionalProcessing was fed under the interpreter (tests/golden/firmware_kat.npz):
    the int16 audio the C program writes is the image's to one count, with the NLMS noise reduction (Init_LMS_NR(15) in
    setup(), nr_level 15 in loop()) and without"""
    kat = np.load(os.path.join(ROOT, "tests", "golden", "firmware_kat.npz"))
    tag, nblk = ("nr15", 48) if nr else ("plain", 32)
    iq = kat["conv_iq"][:nblk * 128]
    fin, fout = tmp_path / "iq.raw", tmp_path / "audio.raw"
    iq.tofile(fin)
    exe = build(tmp_path, defines=["-DRDSP_BIND_LITERAL"], out="binding_literal")
    r = subprocess.run([exe, str(fin), str(fout), str(nblk)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RDSP_NR_LEVEL=str(nr)))
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(fout, dtype=np.int16).reshape(-1, 2)
    want = kat[f"conv_{tag}_o16"]
    assert got.shape == want.shape
    d = np.abs(got.astype(np.int32) - want)
    print(f"C binding, {tag}: {int((d > 0).sum())} of {d.size} samples one count from the image's output")
    assert d.max() <= 1 and (d > 0).sum() <= 400


def test_engine_sketch_binding_compiles(rdsp, tmp_path):
    exe = build(tmp_path, defines=["-DRDSP_BIND_ENGINE"], out="binding_engine")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 64


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sketch_path", "sketch_path_slip"])
def test_the_whole_sketch_over_the_c_binding_against_the_images_audio(rdsp, tmp_path, name):
    """-DRDSP_BIND_ENGINE: setup() and loop() of the sketch in C with `preProcessor.` and `SDR.` bound to the reference's
    own objects (rdsp_preproc_t, rdsp_engine_t) in front of the CONV stage, all of INO:117-139,172-183 as written, fed the
    IQ the firmware image's three routines were fed when chained under the interpreter (tests/golden/sketch_kat.npz):
    the int16 audio the C program plays is the image's to one count"""
    import oracle_lib
    kat = np.load(os.path.join(ROOT, "tests", "golden", "sketch_kat.npz"))
    iq = kat[name + "_iq"]
    nblk = len(iq) // 128
    fin, fout, ftab = tmp_path / "iq.raw", tmp_path / "audio.raw", tmp_path / "tables.raw"
    iq.tofile(fin)
    np.concatenate(oracle_lib.engine_tables()).astype(np.float32).tofile(ftab)
    exe = build(tmp_path, defines=["-DRDSP_BIND_ENGINE"], out="binding_engine")
    r = subprocess.run([exe, str(fin), str(fout), str(nblk)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, RDSP_ENGINE_TABLES=str(ftab)))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "TuningOffset 8390" in r.stdout
    got = np.fromfile(fout, dtype=np.int16).reshape(-1, 2)
    want = kat[name + "_audio"]
    assert got.shape == want.shape
    d = np.abs(got.astype(np.int32) - want)
    print(f"C binding, {name}: {int((d > 0).sum())} of {d.size} samples one count from the image's audio")
    assert d.max() <= 1 and (d > 0).mean() < 0.02


def test_sharding_host_in_c_compiles_and_links(rdsp, tmp_path):
    exe = build(tmp_path, "shard_threads")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 64 and "usage" in r.stderr


@pytest.mark.gpu
def test_channel_shards_on_threads_match_one_bucket_bitwise(rdsp, tmp_path):
    """SURVEY 8(e) in C: one host thread + chain + stream per contiguous channel range (device k % n_devices),
    three shards running concurrently against the same channels as one bucket: K3, pipelined, bit for bit."""
    exe = build(tmp_path, "shard_threads")
    r = subprocess.run([exe, "3", "37", "5", "16"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of" in r.stdout and "3 shards x 37 channels" in r.stdout
