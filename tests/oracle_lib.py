"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and by
the cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")

DEMOD = {"IQ": 0, "USB": 1, "LSB": 2, "CW_USB": 3, "CW_LSB": 4, "AM": 5, "SAM": 6}
AGC = {"off": 0, "fast": 1, "medium": 2, "slow": 3}
ALS = {"off": 0, "notch": 1, "peak": 2}


class OrcConfig(C.Structure):
    _fields_ = [
        ("fs_in", C.c_double),
        ("decim", C.c_int32),
        ("fir_taps", C.c_int32),
        ("fir_cut_hz", C.c_double),
        ("nco_hz", C.c_double),
        ("fft_l", C.c_int32),
        ("window", C.c_int32),
        ("flo_hz", C.c_double),
        ("fhi_hz", C.c_double),
        ("filter_on", C.c_int32),
        ("demod", C.c_int32),
        ("spectral_nr", C.c_int32),
        ("spectral_level", C.c_float),
        ("lms_nr", C.c_int32),
        ("als_mode", C.c_int32),
        ("als_strength", C.c_int32),
        ("agc_mode", C.c_int32),
        ("input_gain", C.c_float),
        ("output_gain", C.c_float),
        ("iq_balance", C.c_float),
        ("mute", C.c_int32),
    ]


DEFAULTS = dict(
    fs_in=96000.0, decim=4, fir_taps=256, fir_cut_hz=10000.0, nco_hz=12000.0,
    fft_l=256, window=1, flo_hz=300.0, fhi_hz=2700.0, filter_on=1, demod="USB",
    spectral_nr=0, spectral_level=0.0, lms_nr=0, als_mode="off", als_strength=20,
    agc_mode="off", input_gain=1.0, output_gain=1.0, iq_balance=1.0, mute=0,
)


def fill_config(struct_cls, **kw):
    """Build a config struct (oracle's or the product's: same field names)."""
    d = dict(DEFAULTS)
    d.update(kw)
    if isinstance(d["demod"], str):
        d["demod"] = DEMOD[d["demod"]]
    if isinstance(d["agc_mode"], str):
        d["agc_mode"] = AGC[d["agc_mode"]]
    if isinstance(d["als_mode"], str):
        d["als_mode"] = ALS[d["als_mode"]]
    s = struct_cls()
    for name, _ in struct_cls._fields_:
        setattr(s, name, d[name])
    return s


def build(native=False, out_dir=None):
    """Compile the oracle.  native=True builds the -O3 -march=native variant used
    only for the cpu_baseline timing leg (built on the box it runs on)."""
    if not native:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("rdsp_oracle.c", "rdsp_oracle.h", "Makefile")]
        # up to date: no child process (a GPU test session may already have initialised the device; tests/conftest.py
        # builds at session start for the same reason)
        if os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(f) for f in srcs):
            return so
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
        return so
    out_dir = out_dir or "/tmp"
    so = os.path.join(out_dir, "liboracle_native.so")
    subprocess.check_call(
        ["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fPIC", "-fopenmp", "-shared",
         "-o", so, os.path.join(ORACLE_DIR, "rdsp_oracle.c"), "-lm"])
    return so


_lib = None


def load(path=None):
    global _lib
    if path is None and _lib is not None:
        return _lib
    # RDSP_ORACLE_SO: another build of the same source (the sanitizer build of test_oracle_kat.py)
    so = path or os.environ.get("RDSP_ORACLE_SO") or os.path.join(ORACLE_DIR, "liboracle.so")
    if not os.path.exists(so):
        build()
    lib = C.CDLL(so)
    f32p = C.POINTER(C.c_float)
    f64p = C.POINTER(C.c_double)
    i16p = C.POINTER(C.c_int16)
    vp = C.c_void_p
    lib.orc_q15_to_float.argtypes = [i16p, f32p, C.c_uint32]
    lib.orc_float_to_q15.argtypes = [f32p, i16p, C.c_uint32]
    lib.orc_cfft_f32.argtypes = [f32p, C.c_uint32, C.c_int]
    lib.orc_calc_cplx_FIR_coeffs.argtypes = [f64p, f64p, C.c_int, C.c_double, C.c_double,
                                             C.c_double, C.c_int]
    lib.orc_init_filter_mask.argtypes = [f32p, f64p, f64p, C.c_uint32]
    lib.orc_chain_create.argtypes = [C.POINTER(OrcConfig)]
    lib.orc_chain_create.restype = vp
    lib.orc_chain_destroy.argtypes = [vp]
    lib.orc_reInitializeFilter.argtypes = [vp, C.c_double, C.c_double]
    lib.orc_doConvolutionalInitialize.argtypes = [vp]
    lib.orc_Init_LMS_NR.argtypes = [vp, C.c_int]
    lib.orc_Init_ALS.argtypes = [vp, C.c_int]
    lib.orc_set_nr_level.argtypes = [vp, C.c_int]
    lib.orc_LMS_NoiseReduction.argtypes = [vp, C.c_int16, f32p]
    lib.orc_chain_process.argtypes = [vp, i16p, C.c_int, i16p, f32p]
    lib.orc_chain_process.restype = C.c_int
    lib.orc_chain_mask.argtypes = [vp]
    lib.orc_chain_mask.restype = f32p
    lib.orc_chain_fir_taps.argtypes = [vp]
    lib.orc_chain_fir_taps.restype = f32p
    lib.orc_chain_lms_coeffs.argtypes = [vp, C.c_int]
    lib.orc_chain_lms_coeffs.restype = f32p
    lib.orc_chain_nfloor.argtypes = [vp]
    lib.orc_chain_nfloor.restype = C.c_float
    lib.orc_chain_agc_gain.argtypes = [vp]
    lib.orc_chain_agc_gain.restype = C.c_float
    lib.orc_chain_nco_dphi.argtypes = [vp]
    lib.orc_chain_nco_dphi.restype = C.c_uint32
    lib.orc_demod_tuning_offset.argtypes = [C.c_int]
    lib.orc_demod_tuning_offset.restype = C.c_uint32
    lib.orc_set_swap_iq.argtypes = [vp, C.c_int]
    lib.orc_set_iq_slip.argtypes = [vp, C.c_int]
    lib.orc_set_noise_blanker.argtypes = [vp, C.c_int, C.c_float]
    lib.orc_set_gains.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_int]
    lib.orc_set_agc_mode.argtypes = [vp, C.c_int]
    lib.orc_set_filter_on.argtypes = [vp, C.c_int]
    lib.orc_set_als_mode.argtypes = [vp, C.c_int]
    lib.orc_set_spectral_nr.argtypes = [vp, C.c_int, C.c_float]
    lib.orc_set_literal_resynthesis.argtypes = [vp, C.c_int]
    lib.orc_set_literal_filter_off.argtypes = [vp, C.c_int]
    lib.orc_set_literal_nr_first_block.argtypes = [vp, C.c_int]
    lib.orc_arm_sin_f32.restype = C.c_float
    lib.orc_arm_sin_f32.argtypes = [C.c_float]
    lib.orc_arm_cos_f32.restype = C.c_float
    lib.orc_arm_cos_f32.argtypes = [C.c_float]
    lib.orc_chain_nb_level.argtypes = [vp]
    lib.orc_chain_nb_level.restype = C.c_float
    lib.orc_set_demod.argtypes = [vp, C.c_int]
    lib.orc_set_nco_hz.argtypes = [vp, C.c_double]
    lib.orc_pbt_step.argtypes = [f64p, f64p, C.c_int, C.c_int]
    lib.orc_passband.argtypes = [C.c_int, C.c_int, f64p, f64p]
    lib.orc_tuning_mode.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.orc_tuning_mode.restype = C.c_int
    lib.orc_multi_process.argtypes = [C.POINTER(OrcConfig), C.c_int, i16p, C.c_int, i16p, C.c_int]
    lib.orc_multi_process.restype = C.c_int
    if path is None:
        _lib = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class OracleChain:
    """One receiver channel run through the oracle."""

    def __init__(self, lib=None, **kw):
        self.lib = lib or load()
        self.cfg = fill_config(OrcConfig, **kw)
        self.h = self.lib.orc_chain_create(C.byref(self.cfg))
        assert self.h
        self.decim = max(1, self.cfg.decim)
        self.fft_l = self.cfg.fft_l

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_chain_destroy(self.h)
            self.h = None

    def process(self, iq):
        """iq: int16 [n_samples, 2] (n_samples multiple of 128).
        Returns (out_i16 [n_out,2], out_f32 [n_out,2])."""
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        n = iq.shape[0]
        assert n % 128 == 0
        nmax = n // self.decim + self.fft_l
        o16 = np.zeros((nmax, 2), np.int16)
        o32 = np.zeros((nmax, 2), np.float32)
        got = self.lib.orc_chain_process(self.h, _p(iq, C.c_int16), n // 128,
                                         _p(o16, C.c_int16), _p(o32, C.c_float))
        return o16[:got].copy(), o32[:got].copy()

    def mask(self):
        return np.ctypeslib.as_array(self.lib.orc_chain_mask(self.h), (2 * self.fft_l,)).copy()

    def fir_taps(self):
        return np.ctypeslib.as_array(self.lib.orc_chain_fir_taps(self.h), (self.cfg.fir_taps,)).copy()

    def lms_coeffs(self, which=0):
        return np.ctypeslib.as_array(self.lib.orc_chain_lms_coeffs(self.h, which), (96,)).copy()

    def nfloor(self):
        return float(self.lib.orc_chain_nfloor(self.h))

    def agc_gain(self):
        return float(self.lib.orc_chain_agc_gain(self.h))

    def reinit_filter(self, lo, hi):
        self.lib.orc_reInitializeFilter(self.h, lo, hi)

    def set_nr_level(self, lvl):
        self.lib.orc_set_nr_level(self.h, lvl)

    def set_swap_iq(self, on):
        self.lib.orc_set_swap_iq(self.h, int(bool(on)))

    def set_iq_slip(self, slip):
        self.lib.orc_set_iq_slip(self.h, int(slip))

    def set_noise_blanker(self, on, threshold_db=10.0):
        self.lib.orc_set_noise_blanker(self.h, int(bool(on)), float(threshold_db))

    def nb_level(self):
        return float(self.lib.orc_chain_nb_level(self.h))

    def set_gains(self, input_gain, iq_balance, output_gain, mute=False):
        self.lib.orc_set_gains(self.h, C.c_float(input_gain), C.c_float(iq_balance), C.c_float(output_gain), int(bool(mute)))

    def set_agc_mode(self, mode):
        self.lib.orc_set_agc_mode(self.h, int(mode))

    def set_filter_on(self, on):
        self.lib.orc_set_filter_on(self.h, int(bool(on)))

    def set_als_mode(self, mode):
        self.lib.orc_set_als_mode(self.h, int(mode))

    def set_spectral_nr(self, on, level):
        self.lib.orc_set_spectral_nr(self.h, int(on), float(level))

    def set_literal_filter_off(self, on):
        """CONV:303 as written: filter off copies only half the spectrum"""
        self.lib.orc_set_literal_filter_off(self.h, int(bool(on)))

    def set_literal_nr_first_block(self, on):
        """CONV:326-337 as written for N_BLOCKS > 1: NR on the first 128 samples of every hop only"""
        self.lib.orc_set_literal_nr_first_block(self.h, int(bool(on)))

    def set_literal_resynthesis(self, on):
        """SPEC:221-235 as written (atan2 + table sin / cos) instead of X * mag'/mag"""
        self.lib.orc_set_literal_resynthesis(self.h, int(bool(on)))

    def set_demod(self, demod):
        self.lib.orc_set_demod(self.h, int(demod))

    def set_nco_hz(self, hz):
        self.lib.orc_set_nco_hz(self.h, float(hz))


def pbt_step(lo, hi, edge, direction, lib=None):
    """checkPBT_Increase/Decrease (CTL:569-612) restated; returns (lo, hi)."""
    lib = lib or load()
    a, b = C.c_double(lo), C.c_double(hi)
    lib.orc_pbt_step(C.byref(a), C.byref(b), edge, direction)
    return a.value, b.value


def passband(filt, demod, lib=None):
    lib = lib or load()
    a, b = C.c_double(), C.c_double()
    lib.orc_passband(filt, demod, C.byref(a), C.byref(b))
    return a.value, b.value


def tuning_mode(mndx, vfo_hz, lib=None):
    """tuningMode() table (CTL:330-423): (ok, filter, demod)."""
    lib = lib or load()
    f, d = C.c_int(), C.c_int()
    ok = lib.orc_tuning_mode(mndx, vfo_hz, C.byref(f), C.byref(d))
    return bool(ok), f.value, d.value


def multi_process(iq, n_threads=1, lib=None, **kw):
    """iq int16 [n_ch, n_samples, 2] -> out int16 [n_ch, n_out, 2] (fresh chains)."""
    lib = lib or load()
    cfg = fill_config(OrcConfig, **kw)
    iq = np.ascontiguousarray(iq, dtype=np.int16)
    n_ch, n, _ = iq.shape
    decim = max(1, cfg.decim)
    out = np.zeros((n_ch, n // decim, 2), np.int16)
    got = lib.orc_multi_process(C.byref(cfg), n_ch, _p(iq, C.c_int16), n // 128,
                                _p(out, C.c_int16), n_threads)
    assert got == n // decim, (got, n // decim)
    return out


# ---- the reference's AudioSDR engine as its firmware image computes it (oracle/rdsp_engine_oracle.c) --------------------
_F32P, _I16P = C.POINTER(C.c_float), C.POINTER(C.c_int16)
ENGINE_TAPS = ["in", "nb", "pre", "mix", "hilbert", "demod", "filt", "agc", "als"]


def _bind_engine(lib):
    if getattr(lib, "_engine_bound", False):
        return lib
    lib.orc_engine_create.restype = C.c_void_p
    lib.orc_engine_create.argtypes = [_F32P, _F32P]
    lib.orc_engine_destroy.argtypes = [C.c_void_p]
    lib.orc_engine_update.argtypes = [C.c_void_p, _I16P, _I16P, _I16P]
    lib.orc_engine_set_tap.argtypes = [C.c_void_p, _F32P]
    lib.orc_engine_setDemodMode.restype = C.c_float
    lib.orc_engine_setDemodMode.argtypes = [C.c_void_p, C.c_int]
    for n in ("setAudioFilter", "setAGCmode", "setMute"):
        getattr(lib, "orc_engine_" + n).argtypes = [C.c_void_p, C.c_int]
    for n in ("setInputGain", "setOutputGain", "setIQgainBalance"):
        getattr(lib, "orc_engine_" + n).argtypes = [C.c_void_p, C.c_float]
    for n in ("enableAGC", "enableAudioFilter", "enableALSfilter", "disableALSfilter", "setALSfilterNotch", "setALSfilterPeak",
              "setALSfilterAdaptive", "disableNoiseBlanker", "enableNoiseBlanker"):
        getattr(lib, "orc_engine_" + n).argtypes = [C.c_void_p]
    lib.orc_engine_scalar.restype = C.c_float
    lib.orc_engine_scalar.argtypes = [C.c_void_p, C.c_int]
    for n in ("agc_curve", "sine", "als_taps"):
        getattr(lib, "orc_engine_" + n).restype = _F32P
        getattr(lib, "orc_engine_" + n).argtypes = [C.c_void_p]
    lib.orc_newlib_expf.restype = C.c_float
    lib.orc_newlib_expf.argtypes = [C.c_float]
    lib._engine_bound = True
    return lib


def engine_tables():
    """the two tables of the engine that have no closed form, as data (tests/golden/firmware_tables.npz)"""
    fw = np.load(os.path.join(_HERE, "golden", "firmware_tables.npz"))
    return np.ascontiguousarray(fw["biquad_sets"].reshape(-1), np.float32), np.ascontiguousarray(fw["hilbert_half64"], np.float32)


class OracleEngine:
    """orc_engine_t: one receiver; `sketch_setup` = INO:120-139"""

    def __init__(self, sketch_setup=True, taps=False):
        self.lib = _bind_engine(load())
        bq, h = engine_tables()
        self.e = self.lib.orc_engine_create(bq.ctypes.data_as(_F32P), h.ctypes.data_as(_F32P))
        self.tapbuf = np.zeros((9, 2, 128), np.float32) if taps else None
        self.taps = {k: [] for k in ENGINE_TAPS}
        if taps:
            self.lib.orc_engine_set_tap(self.e, self.tapbuf.ctypes.data_as(_F32P))
        if sketch_setup:
            for c in (("enableAGC",), ("setAGCmode", 2), ("disableALSfilter",), ("disableNoiseBlanker",), ("setInputGain", 1.0), ("setOutputGain", 0.5),
                      ("setIQgainBalance", 1.02), ("enableAudioFilter",), ("setAudioFilter", 6), ("setDemodMode", 0)):
                self.call(*c)

    def __del__(self):
        if getattr(self, "e", None):
            self.lib.orc_engine_destroy(self.e)
            self.e = None

    def call(self, name, *args):
        return getattr(self.lib, "orc_engine_" + name)(self.e, *args)

    def update(self, i128, q128):
        i, q, o = np.ascontiguousarray(i128, np.int16), np.ascontiguousarray(q128, np.int16), np.zeros(128, np.int16)
        self.lib.orc_engine_update(self.e, i.ctypes.data_as(_I16P), q.ctypes.data_as(_I16P), o.ctypes.data_as(_I16P))
        if self.tapbuf is not None:
            for k, n in enumerate(ENGINE_TAPS):
                self.taps[n].append(self.tapbuf[k].copy())
        return o

    def run(self, iq, calls=()):
        """iq int16 [n, 2]; calls [[block, method, args...]] are made before that block -> int16 [n]"""
        out = np.zeros(len(iq), np.int16)
        for b in range(len(iq) // 128):
            for c in calls:
                if c[0] == b:
                    self.call(c[1], *c[2:])
            out[b * 128:(b + 1) * 128] = self.update(iq[b * 128:(b + 1) * 128, 0], iq[b * 128:(b + 1) * 128, 1])
        return out

    def final(self):
        """the fixture's `_final` row: oscillator phase, AGC gain / envelope / hang / active, PLL Hz / lock, blanker hit"""
        s = lambda k: self.lib.orc_engine_scalar(self.e, k)
        return np.array([s(0), s(1), s(2), s(3), s(4), s(5), s(6), s(9)], np.float32)


class OraclePreProcessor:
    """orc_preproc_t: AudioSDRpreProcessor (INO:53) for one receiver, as the sketch starts it (INO:117)"""

    def __init__(self, start=True, swap=False):
        self.lib = load()
        self.lib.orc_preproc_create.restype = C.c_void_p
        for n, a in (("destroy", []), ("startAutoI2SerrorDetection", []), ("swapIQ", [C.c_int]), ("state", [C.c_int]), ("update", [_I16P, _I16P])):
            getattr(self.lib, "orc_preproc_" + n).argtypes = [C.c_void_p] + a
        self.p = self.lib.orc_preproc_create()
        if start:
            self.lib.orc_preproc_startAutoI2SerrorDetection(self.p)
        if swap:
            self.lib.orc_preproc_swapIQ(self.p, 1)

    def __del__(self):
        if getattr(self, "p", None):
            self.lib.orc_preproc_destroy(self.p)
            self.p = None

    def run(self, iq):
        """int16 [n, 2] -> (int16 [n, 2], state int16 [blocks, 4] = remedy, bad count, counted blocks, detecting)"""
        out, st = np.zeros_like(iq), np.zeros((len(iq) // 128, 4), np.int16)
        for b in range(len(iq) // 128):
            i, q = np.ascontiguousarray(iq[b * 128:(b + 1) * 128, 0]), np.ascontiguousarray(iq[b * 128:(b + 1) * 128, 1])
            self.lib.orc_preproc_update(self.p, i.ctypes.data_as(_I16P), q.ctypes.data_as(_I16P))
            out[b * 128:(b + 1) * 128, 0], out[b * 128:(b + 1) * 128, 1] = i, q
            st[b] = [self.lib.orc_preproc_state(self.p, k) for k in range(4)]
        return out, st
