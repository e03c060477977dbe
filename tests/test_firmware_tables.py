"""The reference-held pins: constant tables cut out of the reference's own shipped firmware image
(tests/golden/firmware_tables.npz, made by tests/golden/make_firmware_tables.py in the build
container) against the generators of the product (host C in librdsp_hip.so, no GPU needed) and of
the oracle.  These are the only numbers in this repository that come from the reference itself."""
import ctypes as C
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
I16P = C.POINTER(C.c_int16)
U16P = C.POINTER(C.c_uint16)


@pytest.fixture(scope="module")
def fw():
    return np.load(os.path.join(HERE, "golden", "firmware_tables.npz"))


@pytest.fixture(scope="module")
def plib(rdsp):
    lib = rdsp.load()
    lib.rdsp_q15_twiddles.argtypes = [C.c_int, C.POINTER(C.c_uint32)]
    lib.rdsp_q15_twiddles.restype = None
    lib.rdsp_sqrt_guess_table.restype = U16P
    return lib


def _olib(oracle):
    lib = oracle.load()
    lib.orc_window_q15_n.argtypes = [C.c_int, C.c_int, I16P]
    lib.orc_twiddle_q15_4096.argtypes = [I16P]
    lib.orc_sqrt_guess_table.restype = U16P
    lib.orc_sqrt_uint32_approx.restype = C.c_uint32
    lib.orc_sqrt_uint32_approx.argtypes = [C.c_uint32]
    lib.orc_sqrt_uint32.restype = C.c_uint32
    lib.orc_sqrt_uint32.argtypes = [C.c_uint32]
    lib.orc_cfft_radix4_q15_n.argtypes = [I16P, C.c_int]
    return lib


PINNED_WINDOWS = [("hann256", 1, 256), ("hann1024", 1, 1024), ("blackman_nuttall256", 3, 256)]


@pytest.mark.parametrize("name,wid,n", PINNED_WINDOWS)
def test_window_tables_equal_the_firmware_image(fw, plib, oracle, name, wid, n):
    """AudioWindowHanning256 (INO:144), AudioWindowHanning1024 (INO:147) and the constructor's default
    AudioWindowBlackmanNuttall256 (FFTIQ.h:56): product and oracle generators reproduce the image's tables."""
    want = fw[name]
    assert want.shape == (n,) and want.dtype == np.int16
    got = np.zeros(n, np.int16)
    plib.rdsp_window_q15_n(wid, n, got.ctypes.data_as(I16P))
    assert np.array_equal(got, want)
    got_o = np.zeros(n, np.int16)
    _olib(oracle).orc_window_q15_n(wid, n, got_o.ctypes.data_as(I16P))
    assert np.array_equal(got_o, want)
    # what rounds 1-4 generated instead, for the record: lround(32767 w(i / N))
    old = np.round(32767 * 0.5 * (1 - np.cos(2 * np.pi * np.arange(n) / n))) if wid == 1 else None
    if old is not None:
        assert np.abs(old - want).max() in (309, 77)


def test_unpinned_windows_are_sane_and_agree_between_product_and_oracle(plib, oracle):
    olib = _olib(oracle)
    for wid in range(1, 12):
        for n in (256, 1024):
            a, b = np.zeros(n, np.int16), np.zeros(n, np.int16)
            plib.rdsp_window_q15_n(wid, n, a.ctypes.data_as(I16P))
            olib.orc_window_q15_n(wid, n, b.ctypes.data_as(I16P))
            assert np.abs(a.astype(int) - b.astype(int)).max() <= 1, wid   # two libm evaluations of the same formula
            assert np.array_equal(a, a[::-1]) or np.abs(a.astype(int) - a[::-1].astype(int)).max() <= 1
            assert a.max() >= 32600 and a.min() >= (-2400 if wid == 6 else 0) and a[0] < 3000   # flat-top has negative side lobes


def test_q15_twiddles_equal_the_firmware_image(fw, plib, oracle):
    """twiddleCoef_4096_q15 (arm_cfft_radix4_q15, FFTIQ.cpp:82): floor(32768 x) clamped; the 256- and
    1024-point plans read it with stride 16 and 4 (arm_cfft_radix4_init_q15)."""
    want = fw["twiddle_q15_4096"]
    t = np.zeros(6144, np.int16)
    _olib(oracle).orc_twiddle_q15_4096(t.ctypes.data_as(I16P))
    assert np.array_equal(t, want)
    for n in (256, 1024):
        tw = np.zeros(3 * n // 4, np.uint32)
        plib.rdsp_q15_twiddles(n, tw.ctypes.data_as(C.POINTER(C.c_uint32)))
        cos = (tw & 0xFFFF).astype(np.uint16).view(np.int16)
        sin = (tw >> 16).astype(np.uint16).view(np.int16)
        step = 4096 // n
        assert np.array_equal(cos, want[0::2][::step][:len(cos)])
        assert np.array_equal(sin, want[1::2][::step][:len(sin)])
        assert sin[:3 * (n // 4 - 1) + 1].min() > -32768     # every entry a plan reads (m <= 3 (n/4 - 1)): the kernels negate it


def test_bit_reversal_walk_with_the_images_table_is_a_bit_reversal(fw):
    """arm_bitreversal_q15 replayed with armBitRevTable from the image: the oracle and the kernels apply the
    permutation directly."""
    tab = fw["bitrev_1024"]
    for n, factor, first in ((256, 16, 15), (1024, 4, 3)):   # pBitRevTable = &armBitRevTable[factor - 1]
        src = list(range(n))
        half = n // 2
        j, k = 0, first
        for i in range(0, half - 1, 2):
            if i < j:
                src[i], src[j] = src[j], src[i]
                src[i + half + 1], src[j + half + 1] = src[j + half + 1], src[i + half + 1]
            src[i + 1], src[j + half] = src[j + half], src[i + 1]
            j = int(tab[k])
            k += factor
        bits = n.bit_length() - 1
        want = [int(format(i, f"0{bits}b")[::-1], 2) for i in range(n)]
        assert src == want


def test_sqrt_guess_table_and_the_literal_routine(fw, plib, oracle):
    """sqrt_uint32_approx (FFTIQ.cpp:105): guess table of the image, two Newton steps; product host twin and
    oracle agree everywhere tried, and the literal sits -1 ... +8 counts from the exact floor root."""
    olib = _olib(oracle)
    want = fw["sqrt_guess"]
    assert np.array_equal(np.ctypeslib.as_array(plib.rdsp_sqrt_guess_table(), (33,)), want)
    assert np.array_equal(np.ctypeslib.as_array(olib.orc_sqrt_guess_table(), (33,)), want)
    rng = np.random.default_rng(3)
    xs = np.unique(np.concatenate([np.arange(0, 70000), (np.arange(1, 65536, 7).astype(np.uint64) ** 2),
                                   rng.integers(0, 2 ** 32, 60000, dtype=np.uint64),
                                   (2.0 ** rng.uniform(0, 32, 40000)).astype(np.uint64),
                                   np.array([2 ** 31, 2 ** 32 - 1], np.uint64)]))
    xs = xs[xs < 2 ** 32]
    lo, hi = 0, 0
    for x in xs:
        x = int(x)
        a = int(olib.orc_sqrt_uint32_approx(x))
        assert a == int(plib.rdsp_sqrt_uint32_approx(x)), x
        n = int(want[32 if x == 0 else 32 - x.bit_length()])    # the routine as published, in Python
        if n:
            n = (x // n + n) // 2
            n = (x // n + n) // 2
        assert a == n, x
        e = int(olib.orc_sqrt_uint32(x))
        lo, hi = min(lo, a - e), max(hi, a - e)
    print(f"sqrt_uint32_approx - floor(sqrt): {lo} ... +{hi}")
    assert lo >= -1 and hi <= 10
    assert int(olib.orc_sqrt_uint32_approx(0)) == 0 and int(olib.orc_sqrt_uint32_approx(3)) == 2


def test_read_range_keeps_the_loop_as_written(plib, oracle):
    """FFTIQ.h:75-86: `do { sum += output[binFirst++]; } while (binFirst < binLast);` never adds binLast
    (unless the two are equal); arguments swap, clamp to 255, bins > 255 read 0."""
    out = (np.arange(256, dtype=np.uint16) * 7 + 3)
    p = out.ctypes.data_as(U16P)
    k = 1.0 / 16384.0
    assert plib.rdsp_spectrum_read(p, 5) == out[5] * k and plib.rdsp_spectrum_read(p, 256) == 0.0
    assert plib.rdsp_spectrum_read_range(p, 10, 14) == float(out[10:14].sum()) * k        # 10..13
    assert plib.rdsp_spectrum_read_range(p, 14, 10) == float(out[10:14].sum()) * k        # swapped
    assert plib.rdsp_spectrum_read_range(p, 7, 7) == out[7] * k                           # the do-while's one pass
    assert plib.rdsp_spectrum_read_range(p, 250, 400) == float(out[250:255].sum()) * k    # clamp to 255, 255 not added
    assert plib.rdsp_spectrum_read_range(p, 300, 400) == 0.0
    # the 1024-point analyser of the Teensy library includes binLast
    o2 = (np.arange(512, dtype=np.uint16) * 3 + 1)
    p2 = o2.ctypes.data_as(U16P)
    assert plib.rdsp_fft1024_read_range(p2, 10, 14) == float(o2[10:15].sum()) * k
    assert plib.rdsp_fft1024_read(p2, 512) == 0.0
    # oracle's restatement of the same lines
    olib = oracle.load()
    olib.orc_fft256iq_create.restype = C.c_void_p
    olib.orc_fft256iq_create.argtypes = [C.c_int, C.c_int]
    olib.orc_fft256iq_read_range.restype = C.c_float
    olib.orc_fft256iq_read_range.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    olib.orc_fft256iq_output.restype = U16P
    olib.orc_fft256iq_output.argtypes = [C.c_void_p]
    olib.orc_fft256iq_destroy.argtypes = [C.c_void_p]
    s = olib.orc_fft256iq_create(1, 0)
    np.ctypeslib.as_array(olib.orc_fft256iq_output(s), (256,))[:] = out
    for a, b in ((10, 14), (14, 10), (7, 7), (250, 400), (300, 400), (0, 255)):
        assert olib.orc_fft256iq_read_range(s, a, b) == plib.rdsp_spectrum_read_range(p, a, b)
    olib.orc_fft256iq_destroy(s)


def test_float_twiddles_of_the_image_are_the_float_cosines(fw):
    """twiddleCoef_256 / _128 (arm_cfft_f32, CONV:291,309): float32 cos / sin of 2 pi k / N to the last bit but
    for entries that are zero in exact arithmetic -- nothing a float FFT restatement could be pinned by beyond
    'its twiddles are correctly rounded', which the oracle's are."""
    for n in (256, 128):
        t = fw[f"twiddle_f32_{n}"].astype(np.float64)
        k = np.arange(n)
        assert np.abs(t[0::2] - np.cos(2 * np.pi * k / n)).max() < 6.1e-8
        assert np.abs(t[1::2] - np.sin(2 * np.pi * k / n)).max() < 6.1e-8


def test_engine_tables_in_the_image_are_what_the_docs_say(fw):
    """The engine's IIR sets (15 x 4 sections), one side of its Hilbert transformer and its 257-entry sine
    table: evidence about the un-vendored AudioSDR library, not pins of this build's chain."""
    bq = fw["biquad_sets"]
    assert bq.shape == (15, 4, 5)
    assert np.allclose(bq[0, 0], [0.28125435, -0.562494, 0.28125435, 1.9529972, -0.95392203], atol=1e-7)
    h = fw["hilbert_half64"].astype(np.float64)
    n = 2 * np.arange(63, -1, -1) + 1                       # odd tap distances 127 ... 1
    assert np.all(h < 0) and np.abs(h[-4:] * n[-4:] * np.pi / 2 + 1).max() < 0.01   # -2 / (pi n) near the centre (windowed further out)
    s = fw["sine257"]
    assert np.array_equal(s, np.sin(2 * np.pi * np.arange(257) / 256).astype(np.float32)) or \
        np.abs(s - np.sin(2 * np.pi * np.arange(257) / 256)).max() < 1e-7


def test_scalar_constants_of_the_image(fw, plib, oracle):
    """What the image says about numbers the sources leave to headers that are not in the tree, or that a restatement
    could get wrong: AUDIO_SAMPLE_RATE_EXACT (CONV:35) is 44100.0 in this build; arm_lms_norm_f32's energy floor is
    1.19209289e-7f; the `x 1.1` of CONV:334 is a double; the design routine's literal pool (CONV:127-185) holds pi,
    0.01 (the centre-tap test), 2 pi, 4 pi, 6 pi and the window coefficients -- the product's and the oracle's taps
    computed with exactly these constants are the taps they return."""
    from cases import CONV_LITERAL
    assert fw["sample_rate"][0] == 44100.0 == CONV_LITERAL["fs_in"]
    assert fw["lms_epsilon"][0] == np.float32(0.000000119209289)
    assert fw["nr_gain"][0] == 1.1 and fw["iq_gain_balance"][0] == np.float32(1.020)
    dc = fw["design_constants"]
    assert dc[0] == np.pi and dc[1] == 0.01 and dc[2] == 2 * np.pi and dc[3] == 4 * np.pi and dc[4] == 6 * np.pi
    rows = {2: (dc[6], dc[5], dc[7], dc[8]), 1: (dc[10], dc[9], dc[11], dc[12])}      # a0, a1, a2, a3 (stored a1 first)
    assert rows[1] == (0.35875, 0.48829, 0.14128, 0.01168) and rows[2] == (0.355768, 0.487396, 0.144232, 0.012604)
    assert (dc[14], dc[13]) == (0.3635819, 0.4891775)
    n, lo, hi, fs = 129, 300.0, 4000.0, float(fw["sample_rate"][0])
    plib.rdsp_calc_cplx_FIR_coeffs.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_double, C.c_double, C.c_double, C.c_int]
    olib = oracle.load()
    olib.orc_calc_cplx_FIR_coeffs.argtypes = plib.rdsp_calc_cplx_FIR_coeffs.argtypes
    for wid, a in rows.items():
        i = np.arange(n, dtype=np.float64)
        x = i - (n - 1) / 2.0
        nfc, nfs = (hi / fs - lo / fs) / 2.0, dc[0] * (hi / fs + lo / fs)
        w = a[0] - a[1] * np.cos(dc[2] * i / (n - 1)) + a[2] * np.cos(dc[3] * i / (n - 1)) - a[3] * np.cos(dc[4] * i / (n - 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            z = np.where(np.abs(x) < dc[1], 2.0 * nfc, np.sin(2 * dc[0] * x * nfc) / (dc[0] * x) * w)
        want_i, want_q = z * np.cos(nfs * x), z * np.sin(nfs * x)
        for lib, fn in ((plib, "rdsp_calc_cplx_FIR_coeffs"), (olib, "orc_calc_cplx_FIR_coeffs")):
            ci, cq = np.zeros(n), np.zeros(n)
            getattr(lib, fn)(ci.ctypes.data_as(C.POINTER(C.c_double)), cq.ctypes.data_as(C.POINTER(C.c_double)), n, lo, hi, fs, wid)
            assert np.abs(ci - want_i).max() <= 1e-15 and np.abs(cq - want_q).max() <= 1e-15, (fn, wid)


def test_engine_iir_sets_against_the_builds_butterworth_design(fw, plib):
    """SURVEY Appendix C / CTL:153-177: the first eight coefficient sets of the image are the engine's audio
    band-passes for fs = 44 117.647 Hz, 150 Hz up to 2.1 / 2.3 / 2.5 / 2.7 / 2.9 / 3.1 / 3.3 / 3.9 kHz.  Each is a
    4th-order high-pass and a 4th-order low-pass with their ZEROS ON THE UNIT CIRCLE (21 and 50 Hz; 2.8 and 5.4
    times the upper edge) and 0.05 dB of pass-band ripple: elliptic-type sections.  rdsp_design_audio_iir (what
    RDSP_AUDIO_KIND_IIR designs for a band) is an 8th-order Butterworth band-pass with the same -3 dB edges and its
    zeros at z = +-1: it has the engine's pass band but 6 dB less attenuation at 1.5 x and 10 dB less at 2 x the upper
    edge, and 9 dB less at 75 Hz.  A host that wants the engine's own sections loads them
    (rdsp_sdr_setAudioIIRCoefficients; GPU parity in tests/test_audio_nodes.py)."""
    from scipy import signal
    fs = float(fw["sample_rate"][0])
    f = np.linspace(1.0, fs / 2, 60000)
    plib.rdsp_design_audio_iir.argtypes = [C.c_double] * 3 + [C.POINTER(C.c_float)]

    def resp(c20):
        c = np.asarray(c20, np.float64).reshape(4, 5)
        sos = np.array([[s[0], s[1], s[2], 1.0, -s[3], -s[4]] for s in c])
        return 20 * np.log10(np.abs(signal.sosfreqz(sos, worN=f, fs=fs)[1]) + 1e-300)

    at = lambda m, x: m[np.argmin(np.abs(f - x))]
    uppers = []
    for i, want_up in enumerate((2100, 2300, 2490, 2680, 2885, 3080, 3295, 3900)):
        c = fw["biquad_sets"][i].astype(np.float64)
        for sec in c:                                         # zeros on the unit circle, poles inside it
            zr, pr = np.roots(sec[:3]), np.roots([1.0, -sec[3], -sec[4]])
            assert np.allclose(np.abs(zr), 1.0, atol=2e-5) and np.all(np.abs(pr) < 0.999)
        zf = sorted(abs(np.angle(np.roots(sec[:3])[0])) * fs / (2 * np.pi) for sec in c)
        m = resp(c)
        band = f[m >= m.max() - 3.0]
        lo, up = band.min(), band.max()
        uppers.append(up)
        assert abs(lo - 150) < 5 and abs(up - want_up) < 12, (i, lo, up)
        assert 15 < zf[0] < 25 and 45 < zf[1] < 55 and 2.5 * up < zf[2] < 3.0 * up and zf[3] > 4.0 * up, (i, zf)
        assert m[(f > 300) & (f < 0.8 * up)].min() > -0.1                      # pass-band ripple
        mine = np.zeros(20, np.float32)
        plib.rdsp_design_audio_iir(150.0, float(up), fs, mine.ctypes.data_as(C.POINTER(C.c_float)))
        mm = resp(mine)
        b2 = f[mm >= mm.max() - 3.0]
        assert abs(b2.min() - lo) < 5 and abs(b2.max() - up) < 5             # the same -3 dB band ...
        d15, d2, d75 = at(mm, 1.5 * up) - at(m, 1.5 * up), at(mm, 2 * up) - m[f > 2 * up].max(), at(mm, 75) - at(m, 75)
        assert 4.5 < d15 < 7.5 and 8.5 < d2 < 12.5 and 7.0 < d75 < 11.0, (i, d15, d2, d75)   # ... a softer stop band
    print("engine band-passes, upper -3 dB edges:", [round(u) for u in uppers])
    # the seven sets behind them are narrower filters whose rows do not all parse as {b0, b1, b2, a1, a2} of a
    # symmetric section (set 8's fourth row, the sign of a1 in set 9's second): kept as found, every section stable
    for c in fw["biquad_sets"][8:].astype(np.float64):
        for sec in c:
            assert np.all(np.abs(np.roots([1.0, -sec[3], -sec[4]])) < 1.0)


def test_code_of_the_image_runs_the_operation_sequences_the_restatements_follow(fw):
    """The image cannot be run here, but it can be read.  tests/golden/make_firmware_tables.py lists where a handful
    of Thumb-2 instruction classes occur in it (class names and offsets only): the DSP-extension parallel arithmetic,
    the dual 16-bit multiplies, CLZ and UDIV.  Their order pins the structure of three restatements:

    * arm_radix4_butterfly_q15 (FFTIQ.cpp:82): one run of exactly 50 such operations = first stage, middle stage,
      last stage of the published DSP-extension routine, operation for operation as oracle/rdsp_oracle.c
      (orc_radix4_butterfly_q15) and csrc/rdsp_q15.h restate it -- inputs >> 2 by two __SHADD16 each, __QADD16 /
      __QSUB16 sums, __SHADD16 for output 0, __SMUAD + __SMUSDX per twiddle product, __QASX / __QSAX in the first
      stage and __SHASX / __SHSAX after it, no multiplies in the last stage -- followed by its inverse twin
      (__SMUSD / __SMUADX, the exchange forms swapped), which the analysers never call;
    * AudioAnalyzeFFT256IQ::update (FFTIQ.cpp:86-105): SMUAD (re^2 + im^2) and UDIV (/ naverage) once per loop, two
      loops, then sqrt_uint32_approx = CLZ (the guess index) and two UDIV (the two Newton steps);
    * AudioAnalyzeFFT1024::update: SMUAD, then CLZ and two UDIV straight away -- no division by naverage, i.e. no
      averaging, as restated (docs/widened_rows.md 6b)."""
    names, offs = [str(n) for n in fw["code_ops_names"]], fw["code_ops_offsets"]
    sh2 = ["SHADD16", "SHADD16"]                                  # T = __SHADD16(__SHADD16(T, 0), 0): the input >> 2
    first = (sh2 + sh2 + ["QADD16", "QSUB16"]                     # ya, yc; R = T + S, S = T - S
             + sh2 + sh2 + ["QADD16", "SHADD16", "QSUB16"]        # yb, yd; T = T + U; y0 = (R + T) >> 1; R = R - T
             + ["SMUAD", "SMUSDX"]                                # co2, si2
             + sh2 + sh2 + ["QSUB16", "QASX", "QSAX"]             # yb, yd again; T = T - U; R, S = the two exchange forms
             + ["SMUAD", "SMUSDX", "SMUAD", "SMUSDX"])            # co1, si1; co3, si3
    middle = (["QADD16", "QSUB16", "QADD16", "SHADD16", "SHADD16", "SHSUB16", "SMUAD", "SMUSDX"]
              + ["QSUB16", "SHASX", "SHSAX", "SMUAD", "SMUSDX", "SMUAD", "SMUSDX"])
    last = ["QADD16", "QADD16", "SHADD16", "QADD16", "SHSUB16", "QSUB16", "QSUB16", "SHSAX", "SHASX"]
    forward = first + middle + last
    swap = {"SMUAD": "SMUSD", "SMUSDX": "SMUADX", "QASX": "QSAX", "QSAX": "QASX", "SHASX": "SHSAX", "SHSAX": "SHASX"}
    inverse = [swap.get(o, o) for o in forward]
    par = [i for i, n in enumerate(names) if n not in ("CLZ", "UDIV", "SMUAD") or n == "SMUAD" and names[max(i - 1, 0)].endswith("16")
           or n == "SMUAD" and i + 1 < len(names) and names[i + 1] == "SMUSDX"]
    seq = [names[i] for i in par]
    assert seq == forward + inverse, seq[:60]                     # nothing else in the image uses these instructions
    assert len(forward) == 50 and offs[par[-1]] - offs[par[0]] < 0x600
    # the same counts the restatement's first stage makes: 12 pre-shift halvings, 1 halving sum, 3 twiddle products
    assert first.count("SHADD16") == 13 and first.count("SMUAD") == 3 and middle.count("SMUAD") == 3 and "SMUAD" not in last
    # the analysers' update(): what surrounds the three remaining SMUADs
    rest = [i for i, n in enumerate(names) if n == "SMUAD" and i not in par]
    assert len(rest) == 3
    a, b, c = rest
    assert names[a:a + 2] == ["SMUAD", "UDIV"] and names[b:b + 5] == ["SMUAD", "UDIV", "CLZ", "UDIV", "UDIV"]     # 256IQ
    assert offs[b + 4] - offs[a] < 0x100
    assert names[c:c + 4] == ["SMUAD", "CLZ", "UDIV", "UDIV"] and offs[c + 3] - offs[c] < 0x20                      # FFT1024


def test_code_of_the_image_runs_arm_lms_norm_f32_in_the_restatements_order_and_unfused(fw):
    """arm_lms_norm_f32 (NR:73) sits in the image between the q15 butterflies and the literal pool that holds its
    epsilon.  Its 28 single-precision operations, in address order, are the published per-sample body with its
    four-times unrolled loops -- and every product is rounded before it is added (VMUL then VADD / VSUB: no VMLA, no
    fused VFMA), which is what the oracle's plain C under -ffp-contract=off computes (oracle/Makefile):
        energy -= x0 * x0; energy += in * in;            VMUL VSUB VMUL VADD
        sum += x[i] * w[i]   (4 per pass + a tail loop)  4 VMUL + 4 VADD, VMUL VADD
        energy + eps; e = d - sum; (e * mu) / (...)      VADD, VSUB, VMUL, VDIV
        w[i] += step * x[i]  (4 per pass + a tail loop)  4 x (VMUL VADD), VMUL VADD
    The CMSIS code of the image as a whole is unfused: its dense float regions hold no fused operation at all."""
    names, offs = np.array([str(n) for n in fw["code_vfp_names"]]), fw["code_vfp_offsets"]
    dsp_n, dsp_o = [str(n) for n in fw["code_ops_names"]], fw["code_ops_offsets"]
    end_q15 = int(max(o for o, n in zip(dsp_o, dsp_n) if n in ("SHASX", "SHSAX")))          # last operation of the inverse butterfly
    eps_at = int(fw["lms_epsilon_offset"])
    assert 0 < eps_at - end_q15 < 0x400
    sel = (offs > end_q15) & (offs < eps_at)
    seq = names[sel].tolist()
    assert len(seq) == 28 and not any(n in ("VFMA", "VFMS", "VFNMA", "VFNMS", "VMLA", "VMLS") for n in seq)
    assert seq[:4] == ["VMUL", "VSUB", "VMUL", "VADD"]                                     # the energy, oldest sample out first
    dot = seq[4:14]
    assert dot.count("VMUL") == 5 and dot.count("VADD") == 5 and dot[-2:] == ["VMUL", "VADD"]
    assert seq[14:18] == ["VADD", "VSUB", "VMUL", "VDIV"]                                   # energy + eps; e; e * mu; the division
    assert seq[18:] == ["VMUL", "VADD"] * 5                                                 # the taps
    # arm_biquad_cascade_df1_f32 (the engine's audio filters): the first dense run behind the LMS code is five samples'
    # worth -- four unrolled and the tail loop -- of 5 products and 4 sums each, unfused (csrc/rdsp_biquad.hip df1_acc)
    after = np.where(offs > eps_at)[0]
    run = [int(after[0])]
    while run[-1] + 1 < len(offs) and offs[run[-1] + 1] - offs[run[-1]] <= 0x30:
        run.append(run[-1] + 1)
    bq = names[run].tolist()
    assert bq.count("VMUL") == 25 and bq.count("VADD") == 20 and len(bq) == 45 and bq[:6] == ["VMUL"] * 5 + ["VADD"]
    fused = np.isin(names, ["VFMA", "VFMS", "VFNMA", "VFNMS"])
    # the CMSIS float routines (transforms, magnitudes, LMS) are the dense float regions of the image: none is fused
    dense = [(lo, ((offs >= lo) & (offs < lo + 0x400)).sum()) for lo in range(0, int(offs.max()), 0x400)]
    for lo, cnt in dense:
        if cnt >= 80 and lo != 0x16400:                     # (0x16400 is newlib's powf -- its L1..L6, cp and ln2 constants sit in
                                                            # the literal pool behind it -- compiled with contraction, like the sketch)
            assert not fused[(offs >= lo) & (offs < lo + 0x400)].any(), hex(lo)


def test_pack_routine_of_the_image_rounds(fw, oracle):
    """CONV:346-347 calls arm_float_to_q15, which CMSIS compiles in one of two variants.  The image holds the one under
    ARM_MATH_ROUNDING: the constants 0.5 and -0.5 are loaded once per loop, and every sample is VMUL (x 32768), VCMP #0,
    VADD (the half), VCVT.S32.F32 (toward zero), SSAT #16 -- four unrolled and a tail loop.  (The other variant has no
    compare and no add.)  Oracle and kernels restate exactly this: round to nearest, halves away from zero."""
    names = [str(n) for n in fw["code_pack_names"]]
    start = names.index("VMOV #0.5")
    seq = names[start:]
    sample = ["VMUL", "VCMP #0", "VADD", "VCVT.S32.F32", "SSAT #16"]
    assert seq == ["VMOV #0.5", "VMOV #-0.5"] + sample * 4 + ["VMOV #0.5", "VMOV #-0.5"] + sample
    lib = oracle.load()
    lib.orc_float_to_q15.argtypes = [C.POINTER(C.c_float), I16P, C.c_uint32]
    x = np.array([0.4, 0.5, 0.6, 1.5, 2.5, -0.4, -0.5, -0.6, -1.5, -2.5, 32766.5, 32767.5, -32768.5], np.float32) / np.float32(32768.0)
    q = np.zeros(len(x), np.int16)
    lib.orc_float_to_q15(x.ctypes.data_as(C.POINTER(C.c_float)), q.ctypes.data_as(I16P), len(x))
    assert q.tolist() == [0, 1, 1, 2, 3, 0, -1, -1, -2, -3, 32767, 32767, -32768]


def test_audio_filter_biquad_of_the_image_is_the_fixed_point_routine(fw, oracle):
    """INO:59-60,155-156 use the Teensy library's AudioFilterBiquad, whose source is not in the reference tree; the image
    holds its update(): two samples per loop turn, each five 32 x 16 multiply-accumulates that keep the top 32 of the 48
    product bits -- bottom / top halves alternating B T B T B, then T B T B T --, `SSAT #16, ASR #14`, the residue kept
    with `UBFX #0, #14`, the two outputs packed with PKHBT.  (Run instead of read: tests/test_firmware_kat.py.)  The
    oracle's one-sample-per-turn restatement is checked against a model that does literally what these instructions
    do, on packed pairs, rails included."""
    names = [str(n) for n in fw["code_fixbq_names"]]
    assert names == (["UBFX #0 #14"] + ["SMLAWB", "SMLAWT", "SMLAWB", "SMLAWT", "SMLAWB"] + ["SSAT #16 ASR #14", "UBFX #0 #14"]
                     + ["SMLAWT", "SMLAWB", "SMLAWT", "SMLAWB", "SMLAWT"] + ["SSAT #16 ASR #14", "PKHBT", "UBFX #0 #14"])
    # and what the image does not hold: CMSIS' 513-entry sinTable_f32 -- src/backup/ (the spectral stage, SPEC:229-232) is
    # "not for normal compilation", so its arm_sin_f32 / arm_cos_f32 are not linked; that row stays unpinned
    assert int(fw["cmsis_sin513_occurrences"]) == 0

    def s32(v):
        v &= 0xFFFFFFFF
        return v - (1 << 32) if v & 0x80000000 else v

    def s16(v):
        v &= 0xFFFF
        return v - 65536 if v & 0x8000 else v

    def smlawb(acc, a, pair): return s32(acc + ((a * s16(pair)) >> 16))
    def smlawt(acc, a, pair): return s32(acc + ((a * s16(pair >> 16)) >> 16))
    def ssat16_asr14(v): return max(-32768, min(32767, v >> 14))

    def model(coef, x):                               # coef as stored (a1, a2 already negated), x int16 [even n]
        b0, b1, b2, a1, a2 = (int(c) for c in coef)
        bprev = aprev = 0
        s = 0
        y = np.zeros(len(x), np.int16)
        for i in range(0, len(x), 2):
            in2 = (int(x[i]) & 0xFFFF) | ((int(x[i + 1]) & 0xFFFF) << 16)
            s = smlawb(s, b0, in2); s = smlawt(s, b1, bprev); s = smlawb(s, b2, bprev)
            s = smlawt(s, a1, aprev); s = smlawb(s, a2, aprev)
            out2 = ssat16_asr14(s); s &= 0x3FFF
            s = smlawt(s, b0, in2); s = smlawb(s, b1, in2); s = smlawt(s, b2, bprev)
            s = smlawb(s, a1, out2 & 0xFFFF); s = smlawt(s, a2, aprev)
            bprev = in2
            hi = ssat16_asr14(s); s &= 0x3FFF
            aprev = (out2 & 0xFFFF) | ((hi & 0xFFFF) << 16)                 # PKHBT
            y[i], y[i + 1] = out2, hi
        return y

    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_audio_nodes import TeensyBiquadOracle
    lib = oracle.load()
    rng = np.random.default_rng(11)
    for trial, (kind, f, q) in enumerate((("highpass", 500.0, 0.5), ("lowpass", 3000.0, 0.707), ("notch", 1000.0, 8.0), ("bandpass", 700.0, 30.0))):
        o = TeensyBiquadOracle(lib)
        o.set(0, kind, f, q)
        coef = [int(v) for v in o.o.coef[0]]
        x = rng.integers(-32768, 32768, 8 * 128).astype(np.int16)          # full-scale noise: the rails are reached
        x[:256] = 32767 if trial % 2 else -32768
        want = model(coef, x)
        got = o.update(x)
        assert np.array_equal(got, want), (kind, int(np.argmax(got != want)))
