"""SURVEY 8f row F3 (rest) and the graph's second analyser:
  * AudioFilterBiquad (`biquad1.setHighpass(0, 500, 0.5)` in front of the panadapter,
    RadioDSP_SDR_RX.ino:58-59,75-78,155-156) and the engine's IIR audio filter bank
    (SDR.setAudioFilter, RDSP_controls.h:153-177) -- csrc/rdsp_biquad.hip;
  * AudioAnalyzeFFT1024 on Q_out_L (RadioDSP_SDR_RX.ino:57,87) -- csrc/rdsp_fft1024.hip.
Both libraries are outside the tree: the oracle states the build-defined arithmetic; the design
formulas are anchored on SciPy (float64, this container only).  CPU tests: design and oracle
known answers.  GPU tests: bit-exact (integer FFT; int16-in/int16-out biquad) or TOL (chain)."""
import ctypes as C

import numpy as np
import pytest

from cases import K1, TOL

F32P, I16P = C.POINTER(C.c_float), C.POINTER(C.c_int16)


def _bind(lib):
    lib.orc_design_butter_bp8.argtypes = [C.c_double, C.c_double, C.c_double, F32P]
    lib.orc_biquad_design.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, F32P]
    lib.orc_cfft_radix4_q15_n.argtypes = [I16P, C.c_int]
    lib.orc_fft1024_create.restype = C.c_void_p
    lib.orc_fft1024_create.argtypes = [C.c_int]
    lib.orc_fft1024_destroy.argtypes = [C.c_void_p]
    lib.orc_fft1024_update.argtypes = [C.c_void_p, I16P]
    lib.orc_fft1024_update.restype = C.c_int
    lib.orc_fft1024_output.argtypes = [C.c_void_p]
    lib.orc_fft1024_output.restype = C.POINTER(C.c_uint16)
    lib.orc_set_audio_iir.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
    lib.orc_chain_iir_coeffs.argtypes = [C.c_void_p]
    lib.orc_chain_iir_coeffs.restype = F32P
    return lib


class OrcBiquad(C.Structure):
    _fields_ = [("n_stages", C.c_int), ("coef", C.c_float * 20), ("state", C.c_float * 16)]


def oracle_biquad(lib, coef20, x):
    """float DF1 cascade of the oracle over a float array (fresh state)"""
    b = OrcBiquad()
    lib.orc_biquad_init.argtypes = [C.POINTER(OrcBiquad), C.c_int, F32P]
    lib.orc_biquad_run.argtypes = [C.POINTER(OrcBiquad), F32P, C.c_int]
    c = np.ascontiguousarray(coef20, np.float32)
    lib.orc_biquad_init(C.byref(b), 4, c.ctypes.data_as(F32P))
    y = np.ascontiguousarray(x, np.float32).copy()
    lib.orc_biquad_run(C.byref(b), y.ctypes.data_as(F32P), len(y))
    return y


class OrcTeensyBiquad(C.Structure):
    _fields_ = [("chained", C.c_int * 4), ("coef", (C.c_int32 * 5) * 4), ("x1", C.c_int16 * 4), ("x2", C.c_int16 * 4),
                ("y1", C.c_int16 * 4), ("y2", C.c_int16 * 4), ("sum", C.c_int32 * 4)]


class TeensyBiquadOracle:
    """the oracle's restatement of the Teensy library's AudioFilterBiquad (fixed point), one channel"""
    KIND = {"lowpass": 0, "highpass": 1, "bandpass": 2, "notch": 3}

    def __init__(self, lib, fs=44100.0):
        self.lib, self.fs, self.o = lib, fs, OrcTeensyBiquad()
        I32P = C.POINTER(C.c_int32)
        lib.orc_teensy_biquad_init.argtypes = [C.POINTER(OrcTeensyBiquad)]
        lib.orc_teensy_biquad_setCoefficients_int.argtypes = [C.POINTER(OrcTeensyBiquad), C.c_int, I32P]
        lib.orc_teensy_biquad_setCoefficients.argtypes = [C.POINTER(OrcTeensyBiquad), C.c_int, C.POINTER(C.c_double)]
        lib.orc_teensy_biquad_design.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, I32P]
        lib.orc_teensy_biquad_update.argtypes = [C.POINTER(OrcTeensyBiquad), I16P, C.c_int]
        lib.orc_teensy_biquad_init(C.byref(self.o))

    def set(self, stage, kind, f, q):
        c5 = np.zeros(5, np.int32)
        self.lib.orc_teensy_biquad_design(self.KIND[kind], f, q, self.fs, c5.ctypes.data_as(C.POINTER(C.c_int32)))
        self.lib.orc_teensy_biquad_setCoefficients_int(C.byref(self.o), stage, c5.ctypes.data_as(C.POINTER(C.c_int32)))
        return c5

    def setCoefficients(self, stage, c5):
        c = np.ascontiguousarray(c5, np.float64)
        self.lib.orc_teensy_biquad_setCoefficients(C.byref(self.o), stage, c.ctypes.data_as(C.POINTER(C.c_double)))

    def update(self, x):
        """x int16 [n] (n a multiple of 128): block by block like the audio interrupt; returns the filtered int16"""
        y = np.ascontiguousarray(x, np.int16).copy()
        for b in range(len(y) // 128):
            blk = y[b * 128:(b + 1) * 128]
            self.lib.orc_teensy_biquad_update(C.byref(self.o), blk.ctypes.data_as(I16P), 128)
        return y


def oracle_fft1024(lib, x, window):
    s = lib.orc_fft1024_create(window)
    outs = []
    for b in range(len(x) // 128):
        blk = np.ascontiguousarray(x[b * 128:(b + 1) * 128], np.int16)
        if lib.orc_fft1024_update(s, blk.ctypes.data_as(I16P)):
            outs.append(np.ctypeslib.as_array(lib.orc_fft1024_output(s), (512,)).copy())
    lib.orc_fft1024_destroy(s)
    return np.stack(outs) if outs else np.zeros((0, 512), np.uint16)


# ---- CPU: designs and oracle known answers ------------------------------------------------------
@pytest.mark.parametrize("f2", [2100.0, 2700.0, 3100.0, 3900.0])
def test_audio_iir_design_is_the_butterworth_bandpass(rdsp, oracle, f2):
    """8th-order band-passes 150 Hz .. 2.1/2.7/3.1/3.9 kHz.  Product design,
    oracle design and SciPy's Butterworth agree; -3 dB sits on the two edges."""
    from scipy import signal
    from radiodsp_sdr_rx_amd.filters import design_audio_iir
    lib = _bind(oracle.load())
    fs = 24000.0
    mine = design_audio_iir(150.0, f2, fs)
    ref = np.zeros(20, np.float32)
    lib.orc_design_butter_bp8(150.0, f2, fs, ref.ctypes.data_as(F32P))
    assert np.abs(mine - ref).max() <= 2e-6          # two independently written designs
    w = np.concatenate([np.linspace(20, 11900, 600), [150.0, f2, np.sqrt(150.0 * f2)]])

    def resp(c):
        sos = np.array([[c[5 * s], c[5 * s + 1], c[5 * s + 2], 1.0, -c[5 * s + 3], -c[5 * s + 4]] for s in range(4)], np.float64)
        return np.abs(signal.sosfreqz(sos, worN=w, fs=fs)[1])

    h = resp(mine)
    h_ref = np.abs(signal.sosfreqz(signal.butter(4, [150.0, f2], btype="band", fs=fs, output="sos"), worN=w, fs=fs)[1])
    assert np.abs(h - h_ref).max() < 1e-4
    assert abs(20 * np.log10(h[-3]) + 3.0103) < 0.01 and abs(20 * np.log10(h[-2]) + 3.0103) < 0.01
    assert abs(h[-1] - 1.0) < 1e-5


@pytest.mark.parametrize("kind,btype", [(0, "lowpass"), (1, "highpass")])
def test_rbj_sections_known_answers(rdsp, oracle, kind, btype):
    """setLowpass / setHighpass at q = 0.7071 are 2nd-order Butterworth sections (RBJ cookbook);
    setBandpass peaks at 1 at the centre; setNotch is zero there; INO:155 (500 Hz, q 0.5)."""
    from scipy import signal
    from radiodsp_sdr_rx_amd.filters import biquad_design
    lib = _bind(oracle.load())
    fs = 44100.0
    c = biquad_design(kind, 1000.0, np.sqrt(0.5), fs)
    o = np.zeros(5, np.float32)
    lib.orc_biquad_design(kind, 1000.0, float(np.sqrt(0.5)), fs, o.ctypes.data_as(F32P))
    assert np.abs(c - o).max() <= 1e-7
    b, a = signal.butter(2, 1000.0, btype=btype, fs=fs)
    assert np.abs(c[:3] - b).max() < 1e-6 and abs(c[3] + a[1]) < 1e-6 and abs(c[4] + a[2]) < 1e-6
    for k, want in ((2, 1.0), (3, 0.0)):
        cc = biquad_design(k, 1000.0, 2.0, fs).astype(np.float64)
        z = np.exp(-2j * np.pi * 1000.0 / fs)
        hc = (cc[0] + cc[1] * z + cc[2] * z * z) / (1.0 - cc[3] * z - cc[4] * z * z)
        assert abs(abs(hc) - want) < 3e-6   # float32 coefficients
    hp = biquad_design(1, 500.0, 0.5, fs).astype(np.float64)   # biquad1.setHighpass(0, 500, 0.5)
    assert abs((hp[0] + hp[1] + hp[2]) / (1.0 - hp[3] - hp[4])) < 1e-6          # no DC
    assert abs(abs((hp[0] - hp[1] + hp[2]) / (1.0 + hp[3] - hp[4])) - 1.0) < 1e-5  # unity at fs/2


def test_oracle_biquad_cascade_matches_scipy(oracle):
    from scipy import signal
    lib = _bind(oracle.load())
    coef = np.zeros(20, np.float32)
    lib.orc_design_butter_bp8(150.0, 2700.0, 24000.0, coef.ctypes.data_as(F32P))
    rng = np.random.default_rng(3)
    x = (0.3 * rng.standard_normal(4096)).astype(np.float32)
    y = oracle_biquad(lib, coef, x)
    c = coef.astype(np.float64)
    sos = np.array([[c[5 * s], c[5 * s + 1], c[5 * s + 2], 1.0, -c[5 * s + 3], -c[5 * s + 4]] for s in range(4)])
    ref = signal.sosfilt(sos, x.astype(np.float64))
    assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-5   # float32 recursion against float64


def test_oracle_fft1024_known_answers(oracle):
    """frames after 8 blocks then every 4; a full-scale-ish tone on bin 37 shows up there with the
    magnitude the 1/1024 scaling of the radix-4 passes gives; Hann spreads it over three bins."""
    lib = _bind(oracle.load())
    n = np.arange(20 * 128)
    x = np.round(16000 * np.cos(2 * np.pi * 37 * n / 1024)).astype(np.int16)
    out = oracle_fft1024(lib, x, 0)
    assert out.shape == (4, 512)                     # 20 blocks: frames at 8, 12, 16, 20
    assert out[0].argmax() == 37 and abs(int(out[0, 37]) - 8000) <= 12
    assert np.delete(out[0], 37).max() <= 12         # fixed-point noise floor
    hann = oracle_fft1024(lib, x, 1)
    assert hann[1].argmax() == 37 and abs(int(hann[1, 37]) - 4000) <= 12
    assert abs(int(hann[1, 36]) - 2000) <= 12 and abs(int(hann[1, 38]) - 2000) <= 12
    # the generic radix-4 routine at 256 points is the F1 routine
    rng = np.random.default_rng(2)
    buf = (rng.standard_normal(512) * 4000).astype(np.int16)
    a, b = buf.copy(), buf.copy()
    lib.orc_cfft_radix4_q15_n(a.ctypes.data_as(I16P), 256)
    lib.orc_cfft_radix4_q15_256.argtypes = [I16P]
    lib.orc_cfft_radix4_q15_256(b.ctypes.data_as(I16P))
    assert np.array_equal(a, b)


# ---- GPU ---------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("window", ["none", "AudioWindowHanning1024", "AudioWindowBlackmanHarris1024", "AudioWindowFlattop1024"])
def test_gpu_fft1024_is_bit_exact(rdsp, oracle, window):
    import torch
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.filters import WINDOWS, AnalyzeFFT1024
    lib = _bind(oracle.load())
    nch, nblk = 5, 44
    audio = synth_iq(nch, nblk * 128)[..., 0].copy()          # any int16 stream will do
    rng = np.random.default_rng(21)                            # ... and rails, to drive the saturating paths
    audio[3] = rng.choice(np.array([-32768, 32767], np.int16), size=audio.shape[1])
    audio[4] = np.repeat(rng.choice(np.array([-32768, 32767], np.int16), size=audio.shape[1] // 128), 128)
    want = [oracle_fft1024(lib, audio[c], WINDOWS[window]) for c in range(nch)]
    dev = torch.from_numpy(audio).cuda()
    an = AnalyzeFFT1024(nch, window=window)
    one = an.update(dev).cpu().numpy().view(np.uint16)
    assert one.shape == (nch, 10, 512)
    for c in range(nch):
        assert np.array_equal(one[c], want[c])
    # split calls (3, 5, 1, 7, 12, 16 blocks): the buffered blocks live on the device
    an2 = AnalyzeFFT1024(nch, window=window)
    got, pos = [], 0
    for k in (3, 5, 1, 7, 12, 16):
        got.append(an2.update(dev[:, pos * 128:(pos + k) * 128]).cpu().numpy().view(np.uint16))
        pos += k
    assert np.array_equal(np.concatenate(got, axis=1), one)
    assert an2.available() and not an2.available()


@pytest.mark.gpu
def test_gpu_fft1024_with_the_firmware_window_by_pointer(rdsp, oracle):
    """`AudioFFT.windowFunction(AudioWindowHanning1024)` (INO:147) with the table of the reference's firmware
    image handed over by pointer, an arbitrary table, and NULL; read(bin) / read(first, last)."""
    import os
    import torch
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.filters import AnalyzeFFT1024
    lib = _bind(oracle.load())
    lib.orc_fft1024_windowFunction_table.argtypes = [C.c_void_p, I16P]
    fw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "firmware_tables.npz"))
    nch, nblk = 3, 28
    audio = synth_iq(nch, nblk * 128)[..., 1].copy()
    dev = torch.from_numpy(audio).cuda()
    for tab in (fw["hann1024"], np.random.default_rng(8).integers(-32768, 32768, 1024).astype(np.int16), None):
        an = AnalyzeFFT1024(nch, window="AudioWindowBlackmanHarris1024")
        an.windowFunction(tab)
        got = an.update(dev).cpu().numpy().view(np.uint16)
        for c in range(nch):
            s = lib.orc_fft1024_create(2)
            lib.orc_fft1024_windowFunction_table(s, tab.ctypes.data_as(I16P) if tab is not None else None)
            want = []
            for b in range(nblk):
                if lib.orc_fft1024_update(s, np.ascontiguousarray(audio[c, b * 128:(b + 1) * 128]).ctypes.data_as(I16P)):
                    want.append(np.ctypeslib.as_array(lib.orc_fft1024_output(s), (512,)).copy())
            lib.orc_fft1024_destroy(s)
            assert np.array_equal(got[c], np.stack(want))
        if tab is fw["hann1024"]:
            named = AnalyzeFFT1024(nch, window="AudioWindowHanning1024").update(dev).cpu().numpy().view(np.uint16)
            assert np.array_equal(named, got)
    row = got[1, -1]
    assert an.read(1, 40) == row[40] / 16384.0 and an.read(1, 512) == 0.0
    assert an.read(1, 30, 34) == float(np.float32(int(row[30:35].sum()))) / 16384.0     # inclusive in the Teensy library


@pytest.mark.gpu
def test_gpu_fft1024_on_the_chain_output_like_the_sketch(rdsp, oracle):
    """`AudioConnection c7(Q_out_L, 0, AudioFFT, 0)` (INO:87): the analyser reads the L side of the
    chain's interleaved int16 audio in place (sample stride 2)."""
    import torch
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.filters import AnalyzeFFT1024
    lib = _bind(oracle.load())
    nch, nblk = 3, 128
    iq = synth_iq(nch, nblk * 128)
    out = Chain(nch, max_blocks_per_call=nblk, **K1).process(torch.from_numpy(iq).cuda())
    torch.cuda.synchronize()
    an = AnalyzeFFT1024(nch, window="AudioWindowHanning1024")
    spec = an.update(out[..., 0]).cpu().numpy().view(np.uint16)
    L = out[..., 0].cpu().numpy()
    for c in range(nch):
        assert np.array_equal(spec[c], oracle_fft1024(lib, L[c], 1))
    # the USB tones of the synthetic input (700 Hz, 1900 Hz, interferer 1000 Hz) at 24 kHz / 1024 per bin
    peak = spec[0, -1].astype(int)
    for f in (700.0, 1000.0, 1900.0):
        k = int(round(f / (24000.0 / 1024)))
        assert peak[k - 1:k + 2].max() > 8 * np.median(peak[:128])


def test_teensy_biquad_restatement_known_answers(rdsp, oracle):
    """The Teensy library's fixed-point AudioFilterBiquad as restated (oracle) and as designed by the product's host
    code: a fresh object passes nothing; setHighpass(0, 500, 0.5) (INO:155) has the coefficients of the RBJ high-pass
    times 2^30 with a1, a2 negated, blocks DC and passes 5 kHz at unit gain to a few counts; a stage is only reached
    through its predecessor's hand-on bit; the error feedback makes the long-run mean of a constant input exact."""
    lib = oracle.load()
    o = TeensyBiquadOracle(lib)
    x = (8000 * np.sin(2 * np.pi * 5000 / 44100.0 * np.arange(40 * 128))).astype(np.int16)
    assert not o.update(x).any()                                   # "by default, the filter will not pass anything"
    c5 = o.set(0, "highpass", 500.0, 0.5)
    plib = rdsp.load()
    plib.rdsp_teensy_biquad_design.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_int32)]
    mine = np.zeros(5, np.int32)
    plib.rdsp_teensy_biquad_design(1, 500.0, 0.5, 44100.0, mine.ctypes.data_as(C.POINTER(C.c_int32)))
    assert np.array_equal(mine, c5)
    w0 = 500.0 * (2.0 * 3.141592654 / 44100.0)
    alpha, cw = np.sin(w0) / (2 * 0.5), np.cos(w0)
    want = np.array([(1 + cw) / 2, -(1 + cw), (1 + cw) / 2, -2 * cw, 1 - alpha]) / (1 + alpha) * 2 ** 30
    assert np.abs(c5 - np.trunc(want)).max() <= 1
    y = o.update(x)
    z = np.exp(-1j * 2 * np.pi * 5000 / 44100.0)
    h = abs((want[0] + want[1] * z + want[2] * z * z) / (2 ** 30 + want[3] * z + want[4] * z * z))   # |H(5 kHz)| = 0.991
    assert abs(int(np.abs(y[20 * 128:]).max()) - 8000 * h) < 12 and 0.98 < h < 1.0
    o2 = TeensyBiquadOracle(lib)
    o2.set(0, "highpass", 500.0, 0.5)
    dc = np.full(60 * 128, 12345, np.int16)
    assert np.abs(o2.update(dc)[40 * 128:]).max() <= 1              # no DC
    o3 = TeensyBiquadOracle(lib)                                     # stage 2 without stage 1: only stage 0 ever runs
    o3.setCoefficients(0, [0.5, 0, 0, 0, 0])
    o3.setCoefficients(2, [0.0, 0, 0, 0, 0])
    assert np.array_equal(o3.update(dc), np.full(len(dc), 6172, np.int16)) or np.abs(o3.update(dc).astype(int) - 6172).max() <= 1
    o3.setCoefficients(1, [1.0 - 2.0 ** -30, 0, 0, 0, 0])           # now 0 -> 1 -> 2, and stage 2 passes nothing
    assert not o3.update(dc)[256:].any()
    # error feedback: the mean of a constant through a gain of 1/3 is exact to a fraction of a count
    o4 = TeensyBiquadOracle(lib)
    o4.setCoefficients(0, [1.0 / 3.0, 0, 0, 0, 0])
    out = o4.update(np.full(300 * 128, 1000, np.int16)).astype(np.float64)
    assert abs(out.mean() - 1000 / 3.0) < 0.01 and set(np.unique(out)) <= {333.0, 334.0}


@pytest.mark.gpu
def test_gpu_biquad_is_bit_exact_and_streams(rdsp, oracle):
    """AudioFilterBiquad on int16 audio, the Teensy library's fixed-point arithmetic: the int16 output equals the
    oracle's restatement; state carries across calls; ragged last wave (21 channels); a stage that is not chained
    to is never run; the rails (saturating outputs) included."""
    import torch
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.filters import FilterBiquad
    lib = _bind(oracle.load())
    nch, nblk = 21, 24
    iq = synth_iq(nch, nblk * 128)
    x = iq[..., 0].copy()
    rng = np.random.default_rng(4)
    x[5] = rng.choice(np.array([-32768, 32767], np.int16), size=x.shape[1])
    x[6] = rng.integers(-32768, 32768, size=x.shape[1]).astype(np.int16)
    bq = FilterBiquad(nch, fs=44100.0)
    dev = torch.from_numpy(x).cuda()
    assert not bq.update(dev[:, :128].contiguous()).cpu().numpy().any()        # a fresh object passes nothing
    bq = FilterBiquad(nch, fs=44100.0)
    bq.setHighpass(0, 500, 0.5)                      # INO:155
    bq.setLowpass(1, 3000, 0.7071)
    bq.setNotch(3, 1000, 4.0)                        # stage 2 was never set: stage 3 is never reached
    defn, n_stages = bq.definition()
    assert n_stages == 2 and not defn[2].any() and defn[3].any()
    y = torch.cat([bq.update(dev[:, :5 * 128]), bq.update(dev[:, 5 * 128:])], dim=1).cpu().numpy()
    for c in range(nch):
        o = TeensyBiquadOracle(lib)
        o.set(0, "highpass", 500.0, 0.5)
        o.set(1, "lowpass", 3000.0, 0.7071)
        o.set(3, "notch", 1000.0, 4.0)
        assert np.array_equal(np.asarray(o.o.coef, np.int32).reshape(4, 5), defn)
        assert np.array_equal(y[c], o.update(x[c])), c
    # the Q side of the interleaved stream through a second object (biquad2 of the sketch)
    bq2 = FilterBiquad(nch, fs=44100.0)
    bq2.setHighpass(0, 500, 0.5)
    iqd = torch.from_numpy(iq).cuda()
    yq = bq2.update(iqd[..., 1]).cpu().numpy()
    o = TeensyBiquadOracle(lib)
    o.set(0, "highpass", 500.0, 0.5)
    assert np.array_equal(yq[3], o.update(np.ascontiguousarray(iq[3, :, 1])))
    # four chained stages with explicit double coefficients (setCoefficients(stage, const double *))
    bq3 = FilterBiquad(nch)
    o = TeensyBiquadOracle(lib)
    for st, c5 in enumerate(([0.9, -1.7, 0.9, -1.9, 0.95], [0.2, 0.4, 0.2, -0.3, 0.1], [1.0, 0.0, -1.0, -1.2, 0.7], [0.5, 0.5, 0.0, 0.3, 0.0])):
        bq3.setCoefficients(st, c5)
        o.setCoefficients(st, c5)
    assert bq3.definition()[1] == 4
    assert np.array_equal(bq3.update(dev).cpu().numpy()[5], o.update(x[5]))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_gpu_biquad_random_sessions_are_bit_exact(rdsp, oracle, seed):
    """update() calls of random length with setLowpass / setHighpass / setBandpass / setNotch on random
    stages in between: a new section takes over with the next sample, inherits the stage's sample history and
    starts from a cleared residue, and a stage runs only once every stage in front of it was set, as in the
    library; int16 in, int16 out, identical."""
    import torch
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.filters import FilterBiquad
    lib = _bind(oracle.load())
    rng = np.random.default_rng(seed)
    nch, fs = 5, 44100.0
    ops = []
    for _ in range(12):
        if rng.integers(0, 2):
            ops.append(("update", int(rng.integers(1, 12))))
        else:
            ops.append(("set", int(rng.integers(0, 4)), str(rng.choice(["lowpass", "highpass", "bandpass", "notch"])),
                        float(rng.choice([300.0, 500.0, 1000.0, 3000.0])), float(rng.choice([0.5, 0.7071, 4.0]))))
    ops.append(("update", 6))
    total = sum(op[1] for op in ops if op[0] == "update")
    x = synth_iq(nch, total * 128)[..., 0].copy()
    x[2] = rng.integers(-32768, 32768, size=x.shape[1]).astype(np.int16)
    bq = FilterBiquad(nch, fs=fs)
    ors = [TeensyBiquadOracle(lib, fs) for _ in range(nch)]
    dev = torch.from_numpy(x).cuda()
    setters = {"lowpass": bq.setLowpass, "highpass": bq.setHighpass, "bandpass": bq.setBandpass, "notch": bq.setNotch}
    pos = 0
    for op in ops:
        if op[0] == "update":
            n = op[1] * 128
            got = bq.update(dev[:, pos:pos + n].contiguous()).cpu().numpy()
            for c in range(nch):
                assert np.array_equal(got[c], ors[c].update(x[c, pos:pos + n])), (seed, op, c)
            pos += n
        else:
            _, stage, kind, f, q = op
            setters[kind](stage, f, q)          # the stage's residue goes, its sample history stays, stage - 1 hands on to it
            for o in ors:
                o.set(stage, kind, f, q)


@pytest.mark.gpu
def test_iqinput_biquad_fft_wiring_of_the_sketch(rdsp, oracle):
    """INO:75-78: IQinput -> biquad1 / biquad2 (high-pass 500 Hz) -> FFT (AudioAnalyzeFFT256IQ):
    the panadapter path as graph nodes, against the oracle's biquad + analyser restatements."""
    import torch
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.filters import FilterBiquad
    from radiodsp_sdr_rx_amd.graph import Graph
    from radiodsp_sdr_rx_amd.spectrum import AnalyzeFFT256IQ
    from test_spectrum import _olib, oracle_spectra
    lib = _bind(_olib(oracle))
    nch, nblk = 2, 40
    iq = synth_iq(nch, nblk * 128)
    g = Graph(nch)
    g.AudioMemory(40)                                               # INO:151
    IQinput = g.input_node()                                        # INO:52
    b1, b2 = FilterBiquad(nch), FilterBiquad(nch)                   # INO:58-59
    b1.setHighpass(0, 500, 0.5); b2.setHighpass(0, 500, 0.5)        # INO:155-156
    biquad1, biquad2 = g.biquad_node(b1), g.biquad_node(b2)
    fft = AnalyzeFFT256IQ(nch, naverage=8, window="AudioWindowHanning256")
    FFT = g.spectrum_node(fft)                                      # INO:57
    g.AudioConnection(IQinput, 0, biquad1, 0)                       # INO:75
    g.AudioConnection(IQinput, 1, biquad2, 0)                       # INO:76
    g.AudioConnection(biquad1, 0, FFT, 0)                           # INO:77
    g.AudioConnection(biquad2, 0, FFT, 1)                           # INO:78
    spectra = []
    for b in range(nblk):
        blk = np.ascontiguousarray(iq[:, b * 128:(b + 1) * 128])
        IQinput.push(np.ascontiguousarray(blk[..., 0]), np.ascontiguousarray(blk[..., 1]))
        assert g.update_all() == 0
        if FFT.available():
            spectra.append(FFT.output())
    assert biquad1.status() == 0 and biquad2.status() == 0 and FFT.status() == 0 and len(spectra) == 4
    for c in range(nch):
        filt = np.zeros((nblk * 128, 2), np.int16)
        for side in (0, 1):
            o = TeensyBiquadOracle(lib)
            o.set(0, "highpass", 500.0, 0.5)
            filt[:, side] = o.update(np.ascontiguousarray(iq[c, :, side]))
        want = np.stack(oracle_spectra(lib, filt, 8, 1))
        assert np.array_equal(np.stack([s[c] for s in spectra]), want)


IIR_CASES = {
    "usb_2700": (dict(fft_l=256, demod="USB"), 2),               # audio2700
    "lsb_2100_agc": (dict(fft_l=512, demod="LSB", agc_mode="medium"), 1),
    "cw_agc": (dict(fft_l=256, demod="CW_USB", agc_mode="fast", output_gain=0.5), 0),           # audioCW
    "am_3900": (dict(fft_l=512, demod="AM"), 4),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(IIR_CASES))
def test_engine_audio_filter_as_iir_bank_matches_oracle(rdsp, oracle, name):
    """rdsp_sdr_setAudioFilterKind(IIR): the mask keeps the side band only, SDR.setAudioFilter picks
    an 8th-order band-pass of four biquads on the demodulated audio.  Oracle: same mask band, same
    float coefficients (read back from the chain; the designs are compared in the CPU test)."""
    import torch
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    lib = _bind(oracle.load())
    lib.orc_biquad_init.argtypes = [C.c_void_p, C.c_int, F32P]
    cfg, filt = IIR_CASES[name]
    nch, nblk = 19, 64
    iq = synth_iq(nch, nblk * 128, cw=name.startswith("cw"))
    if name.startswith("cw"):
        cfg = dict(cfg, nco_hz=12000.0)
    ch = Chain(nch, max_blocks_per_call=nblk // 2, **cfg)
    ch.setAudioFilterKind(1)
    ch.setAudioFilter(filt)
    ch.set_pipelined(True)
    outs = [ch.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * (nblk // 2) * 128:(k + 1) * (nblk // 2) * 128])).cuda(),
                       want_f32=True) for k in range(2)]
    ch.flush()
    torch.cuda.synchronize()
    got = np.concatenate([o[1].cpu().numpy() for o in outs], axis=1)
    coef = ch.iir_coeffs(0)
    lo, hi = oracle.passband(filt, rdsp.DEMOD[cfg["demod"]])
    f1, f2 = sorted((abs(lo), abs(hi)))
    if cfg["demod"] == "AM":
        f1, band = 150.0, (-4000.0, 4000.0)
    else:
        band = (-4000.0, -50.0) if cfg["demod"] == "LSB" else (50.0, 4000.0)
    from radiodsp_sdr_rx_amd.filters import design_audio_iir
    assert np.array_equal(coef, design_audio_iir(f1, f2, 24000.0))
    import np_model
    from parity_util import assert_truth_anchored
    ocfg = dict(cfg, flo_hz=band[0], fhi_hz=band[1])
    ref, f64 = [], []
    for c in range(nch):
        oc = oracle.OracleChain(**ocfg)
        lib.orc_set_audio_iir(oc.h, 1, f1, f2)
        # the chain's own float coefficients, so that the comparison is about the arithmetic
        dst = np.ctypeslib.as_array(lib.orc_chain_iir_coeffs(oc.h), (20,))
        dst[:] = coef
        ref.append(oc.process(iq[c])[1])
        f64.append(np_model.Model(iir_coef=coef, **ocfg).process(iq[c]))
    # a recursive filter behind the chain (and, for AM, one that removes most of the envelope's
    # amplitude): truth-anchored like the NLMS chains -- err(gpu, f64) <= max(TOL, 1.5 err(oracle, f64))
    assert_truth_anchored(got, np.stack(ref), np.stack(f64), name)


@pytest.mark.gpu
def test_engine_iir_sets_of_the_firmware_image_run_through_the_product(rdsp, oracle):
    """The reference's firmware image holds the engine's own audio filters (CTL:153-177):
    fifteen sets of four {b0, b1, b2, a1, a2} sections for fs = 44 117.647 Hz.  Every set, as found:
      * through AudioFilterBiquad (setCoefficients per stage; the Teensy convention has a1, a2 in the denominator,
        the table's CMSIS order has them added) on int16 audio -- the Teensy library's fixed-point cascade, every
        coefficient of the sets fits its 2.30 format: bit-exact against the oracle's restatement;
      * the eight band-pass sets through the chain at the reference's native rate (decim 1, fs 44 117.647, USB,
        RDSP_AUDIO_KIND_IIR with the set loaded by rdsp_sdr_setAudioIIRCoefficients), pipelined, two calls:
        truth-anchored against the float64 model like every chain that ends in a recursion."""
    import os
    import torch
    import np_model
    from cases import CONV_LITERAL
    from cases import TOL
    from parity_util import normwise
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.filters import FilterBiquad
    lib = _bind(oracle.load())
    lib.orc_float_to_q15.argtypes = [F32P, I16P, C.c_uint32]
    fw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "firmware_tables.npz"))
    sets = fw["biquad_sets"]
    fs = 44100.0
    nch, nblk = 5, 48
    iq = synth_iq(nch, nblk * 128)
    audio = np.ascontiguousarray(iq[..., 0])
    dev = torch.from_numpy(audio).cuda()
    for i in range(15):
        bq = FilterBiquad(nch, fs=fs)
        o = [TeensyBiquadOracle(lib, fs) for _ in range(nch)]
        for st in range(4):
            b0, b1, b2, a1, a2 = (float(v) for v in sets[i, st])
            bq.setCoefficients(st, [b0, b1, b2, -a1, -a2])
            for oc in o:
                oc.setCoefficients(st, [b0, b1, b2, -a1, -a2])
        got = np.concatenate([bq.update(dev[:, :16 * 128].contiguous()).cpu().numpy(), bq.update(dev[:, 16 * 128:].contiguous()).cpu().numpy()], 1)
        for c in range(nch):
            assert np.array_equal(got[c], o[c].update(audio[c])), (i, c)
    cfg = dict(CONV_LITERAL, demod="USB", flo_hz=50.0, fhi_hz=4000.0, agc_mode="medium")
    devq = torch.from_numpy(iq).cuda()
    for i in range(8):
        coef = sets[i].reshape(-1).astype(np.float32)
        ch = Chain(nch, max_blocks_per_call=nblk // 2, **cfg)
        ch.setAudioFilterKind(1)
        ch.setAudioIIRCoefficients(coef)
        ch.set_pipelined(True)
        outs = [ch.process(devq[:, k * (nblk // 2) * 128:(k + 1) * (nblk // 2) * 128].contiguous(), want_f32=True) for k in range(2)]
        ch.flush()
        torch.cuda.synchronize()
        got = np.concatenate([o[1].cpu().numpy() for o in outs], axis=1)
        assert np.array_equal(ch.iir_coeffs(0), coef)
        ref, f64 = [], []
        for c in range(nch):
            oc = oracle.OracleChain(**cfg)
            lib.orc_set_audio_iir(oc.h, 1, 150.0, 2700.0)
            np.ctypeslib.as_array(lib.orc_chain_iir_coeffs(oc.h), (20,))[:] = coef
            ref.append(oc.process(iq[c])[1])
            f64.append(np_model.Model(iir_coef=coef, **cfg).process(iq[c]))
        # sections with poles at r = 0.995 (the 21 Hz / 150 Hz high-pass pair) turn a last-bit difference of their INPUT
        # into ~200 of them at the output: the cascade's arithmetic is bit-exact on identical input (above), but the
        # front end in front of it rounds differently on the two sides, so the yardstick here is 3 x (not 1.5 x) the
        # oracle's own distance from the float64 evaluation -- both float32 results are draws of the same noise
        f64a, refa = np.stack(f64), np.stack(ref)
        den = np.abs(f64a).max(axis=(1, 2))
        eg = np.abs(got - f64a).max(axis=(1, 2)) / den
        eo = np.abs(refa - f64a).max(axis=(1, 2)) / den
        print(f"firmware IIR set {i}: err(gpu, f64) {eg.max():.2e}, err(oracle, f64) {eo.max():.2e}, gpu vs oracle {normwise(got, refa):.2e}")
        assert eg.max() <= max(TOL, 3.0 * eo.max()) and np.median(eg) <= max(TOL, 2.0 * np.median(eo)), (i, eg, eo)
    # the loader needs the IIR kind and sane numbers
    from radiodsp_sdr_rx_amd import RdspError
    plain = Chain(2, max_blocks_per_call=8, **cfg)
    with pytest.raises(RdspError) as e:
        plain.setAudioIIRCoefficients(sets[0].reshape(-1))
    assert e.value.code == -5
    plain.setAudioFilterKind(1)
    bad = sets[0].reshape(-1).copy()
    bad[7] = np.nan
    with pytest.raises(RdspError):
        plain.setAudioIIRCoefficients(bad)


@pytest.mark.gpu
def test_argument_errors_are_loud(rdsp):
    """the new entry points reject what they cannot do, with RDSP_ERR_* and a message"""
    import ctypes as C
    import torch
    from radiodsp_sdr_rx_amd import RdspError
    from radiodsp_sdr_rx_amd.chain import Chain
    from radiodsp_sdr_rx_amd.filters import AnalyzeFFT1024, FilterBiquad
    lib = rdsp.load()
    bq = FilterBiquad(3)
    with pytest.raises(RdspError):
        bq.setHighpass(4, 500, 0.5)                     # AudioFilterBiquad has stages 0..3
    with pytest.raises(RdspError):
        bq.setLowpass(0, -1.0, 0.7)
    x = torch.zeros((3, 256), dtype=torch.int16, device="cuda")
    out = torch.zeros_like(x)
    assert lib.rdsp_biquad_update(bq.h, C.c_void_p(x.data_ptr()), 128, 1, 2, C.c_void_p(out.data_ptr()), 256, 1, None) == -1
    an = AnalyzeFFT1024(3)
    spectra = torch.zeros((3, 1, 512), dtype=torch.int16, device="cuda")
    got = C.c_int()
    audio = torch.zeros((3, 16 * 128), dtype=torch.int16, device="cuda")   # 16 blocks complete 3 frames
    assert lib.rdsp_fft1024_update(an.h, C.c_void_p(audio.data_ptr()), 16 * 128, 1, 16, C.c_void_p(spectra.data_ptr()), 1,
                                   C.byref(got), None) == -1
    assert b"3 needed" in lib.rdsp_last_error()
    ch = Chain(2, max_blocks_per_call=8, decim=1, fs_in=24000.0, nco_hz=0.0, fft_l=256, demod="IQ")
    with pytest.raises(RdspError) as e:
        ch.set_fir_variant(2)                           # no decimator, no frequency-domain decimator
    assert e.value.code == -5
    with pytest.raises(RdspError):
        ch.setAudioFilterKind(7)
    with pytest.raises(RdspError):
        ch.set_spectral_nr(3, 1.0)
    with pytest.raises(RdspError) as e:
        ch.set_fir_variant(5)                           # ... nor its row form
    assert e.value.code == -5
    with pytest.raises(RdspError):
        ch.set_fir_variant(7)
    if not lib.rdsp_experimental_build():
        with pytest.raises(RdspError) as e:
            ch.set_fir_variant(1)                       # matrix-core FIR: EXPERIMENTAL builds only
        assert e.value.code == -5
        d4 = Chain(2, max_blocks_per_call=8, fft_l=256, demod="USB")
        with pytest.raises(RdspError) as e:
            d4.set_fir_variant(6)                       # the row form with 192 outputs per window: measured, not adopted
        assert e.value.code == -5
        d4.set_fir_variant(5)
    # round 5's entry points
    h = C.c_void_p()
    assert lib.rdsp_spectrum_create(2, 0, 8, 99, C.byref(h)) == -1 and b"window id" in lib.rdsp_last_error()
    assert lib.rdsp_spectrum_create(2, 0, 8, -3, C.byref(h)) == -1
    assert lib.rdsp_fft1024_create(2, 0, 12, C.byref(h)) != 0
    from radiodsp_sdr_rx_amd.spectrum import AnalyzeFFT256IQ
    fft = AnalyzeFFT256IQ(2)
    assert lib.rdsp_spectrum_windowFunction(fft.h, 12) == -1 and lib.rdsp_spectrum_windowFunction_table(None, None) == -1
    assert lib.rdsp_fft1024_windowFunction(an.h, -1) == -1 and lib.rdsp_fft1024_windowFunction_table(None, None) == -1
    assert lib.rdsp_spectrum_windowFunction_table(fft.h, None) == 0           # windowFunction(NULL): no window, FFTIQ.cpp:81
    assert lib.rdsp_set_spectral_resynthesis(None, 1) == -1 and lib.rdsp_set_nlms_energy_mode(None, 1) == -1
    assert lib.rdsp_sdr_setAudioIIRCoefficients(ch.h, None) == -1
    assert lib.rdsp_chain_call_unit_blocks(None) == 0 and lib.rdsp_chain_granule_blocks(None) == 0
    assert ch.call_unit_blocks == 2 and ch.granule_blocks == 2               # decim 1, FFT_L 256: one 256-sample chunk
    with pytest.raises(RdspError) as e:
        ch.process(torch.zeros((2, 3 * 128, 2), dtype=torch.int16, device="cuda"))
    assert e.value.code == -4 and "call unit" in str(e.value)


# ---- committed fixture of the graph nodes (tests/golden/nodes.npz, made by make_golden.py) --------
def _nodes_fixture():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nodes.npz"))


def test_oracle_reproduces_the_node_fixture(oracle):
    """pins the integer analysers and the biquad cascade of the oracle against accidental change"""
    from test_spectrum import _olib, oracle_spectra
    lib = _bind(_olib(oracle))
    lib.orc_biquad_set_stage.argtypes = [C.POINTER(OrcBiquad), C.c_int, F32P]
    lib.orc_biquad_init.argtypes = [C.POINTER(OrcBiquad), C.c_int, F32P]
    lib.orc_biquad_run.argtypes = [C.POINTER(OrcBiquad), F32P, C.c_int]
    lib.orc_float_to_q15.argtypes = [F32P, I16P, C.c_uint32]
    g = _nodes_fixture()
    iq = g["iq"]
    for c in range(iq.shape[0]):
        assert np.array_equal(np.stack(oracle_spectra(lib, iq[c], 5, 1)), g["spectrum256"][c])
        assert np.array_equal(oracle_fft1024(lib, np.ascontiguousarray(iq[c, :, 0]), 1), g["fft1024"][c])
        o = TeensyBiquadOracle(lib)
        o.set(0, "highpass", 500.0, 0.5)
        o.set(1, "notch", 1000.0, 4.0)
        assert np.array_equal(o.update(np.ascontiguousarray(iq[c, :, 0])), g["biquad"][c])


@pytest.mark.gpu
def test_gpu_nodes_match_the_fixture(rdsp):
    """the three node kernels against the committed outputs, bit for bit (no oracle at run time)"""
    import torch
    from radiodsp_sdr_rx_amd.filters import AnalyzeFFT1024, FilterBiquad
    from radiodsp_sdr_rx_amd.spectrum import AnalyzeFFT256IQ
    g = _nodes_fixture()
    iq = g["iq"]
    nch = iq.shape[0]
    dev = torch.from_numpy(iq).cuda()
    s256 = AnalyzeFFT256IQ(nch, naverage=5, window="AudioWindowHanning256").update(dev).cpu().numpy().view(np.uint16)
    assert np.array_equal(s256, g["spectrum256"])
    s1024 = AnalyzeFFT1024(nch, window="AudioWindowHanning1024").update(dev[..., 0]).cpu().numpy().view(np.uint16)
    assert np.array_equal(s1024, g["fft1024"])
    bq = FilterBiquad(nch, fs=44100.0)
    bq.setHighpass(0, 500, 0.5)
    bq.setNotch(1, 1000, 4.0)
    assert np.array_equal(bq.update(dev[..., 0]).cpu().numpy(), g["biquad"])
