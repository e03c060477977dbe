"""Host-side mirror of the reference's stage interface for the receive path.

`Chain` wraps one rdsp_chain_t (n_channels receivers on one GPU).  Method names
follow the reference's free functions (RDSP_convolutional.h:187-228,
RDSP_noise_reduction.h:35) and the AudioSDR setters visible at
RadioDSP_SDR_RX.ino:117-139.  PyTorch is used only for device memory and
streams; the compute is librdsp_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import make_config, synth_config


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


class Chain:
    # stage A3 of chains made without an explicit choice: None = the library's default (frequency domain with
    # frames of one granule: the bits do not depend on the call split); 0 = the direct form; 2 = the
    # frequency-domain decimator with 448-sample frames (what bench.py runs).  The GPU test modules run under
    # all three (tests/conftest.py `front_form`); the default itself is one of two kernels (set_fir_variant).
    default_fir_variant = None

    def __init__(self, n_channels, max_blocks_per_call=512, device=0, fir_variant=None, **cfg):
        self.lib = _lib.load()
        self.cfg = make_config(**cfg)
        self.n_channels = int(n_channels)
        self.device = int(device)
        self.decim = max(1, self.cfg.decim)
        self.fft_l = self.cfg.fft_l
        h = C.c_void_p()
        _lib.check(self.lib.rdsp_chain_create(C.byref(self.cfg), self.n_channels, self.device,
                                              int(max_blocks_per_call), C.byref(h)))
        self.h = h
        if fir_variant is None and self.decim == 4:
            fir_variant = self.default_fir_variant
        if fir_variant is not None:
            self.set_fir_variant(fir_variant)

    @property
    def call_unit_blocks(self):
        """128-sample input blocks per call must be a multiple of this (N_BLOCKS of CONV:38-39 at the input rate)"""
        return self.lib.rdsp_chain_call_unit_blocks(self.h)

    @property
    def granule_blocks(self):
        """cut a stream in multiples of this and the bits do not depend on the cut: the call unit, but with 448-sample
        decimator frames (fir_variant 2) lcm(14, call unit) -- whole frames"""
        return self.lib.rdsp_chain_granule_blocks(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rdsp_chain_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the hot path ------------------------------------------------------
    def process(self, iq, out=None, out_f32=None, want_f32=False, stream=None):
        """iq: int16 cuda tensor [n_channels, n_samples, 2].  Returns int16
        [n_channels, n_samples/decim, 2] (and the float pairs if requested)."""
        assert iq.is_cuda and iq.dtype == torch.int16 and iq.dim() == 3 and iq.shape[2] == 2
        # rows may be slices of a longer buffer: [ch][in_stride][2] with the sample pairs contiguous
        assert iq.shape[0] == self.n_channels and iq.stride(2) == 1 and iq.stride(1) == 2
        n = iq.shape[1]
        assert n % 128 == 0
        n_out = n // self.decim
        if out is None:
            out = torch.empty((self.n_channels, n_out, 2), dtype=torch.int16, device=iq.device)
        if want_f32 and out_f32 is None:
            out_f32 = torch.empty((self.n_channels, n_out, 2), dtype=torch.float32, device=iq.device)
        if out_f32 is not None:
            assert out_f32.stride(0) == out.stride(0)  # one out_stride for both (samples)
        f32p = C.c_void_p(out_f32.data_ptr()) if out_f32 is not None else C.c_void_p(0)
        _lib.check(self.lib.rdsp_chain_process(
            self.h, C.c_void_p(iq.data_ptr()), iq.stride(0) // 2, n // 128,
            C.c_void_p(out.data_ptr()), out.stride(0) // 2, f32p, _stream_ptr(stream)))
        return (out, out_f32) if want_f32 else out

    # ---- reference-named stage calls ----------------------------------------
    def doConvolutionalInitialize(self, stream=None):
        _lib.check(self.lib.rdsp_doConvolutionalInitialize(self.h, _stream_ptr(stream)))

    def reInitializeFilter(self, dFLoCut, dFHiCut, stream=None):
        _lib.check(self.lib.rdsp_reInitializeFilter(self.h, dFLoCut, dFHiCut, _stream_ptr(stream)))

    def Init_LMS_NR(self, strength, stream=None):
        _lib.check(self.lib.rdsp_Init_LMS_NR(self.h, int(strength), _stream_ptr(stream)))

    def LMS_NoiseReduction(self, nrbuffer, stream=None):
        """NR:66 in isolation: float32 cuda tensor [n_channels, n] processed in place."""
        assert nrbuffer.is_cuda and nrbuffer.dtype == torch.float32 and nrbuffer.is_contiguous()
        _lib.check(self.lib.rdsp_LMS_NoiseReduction(self.h, nrbuffer.shape[1], C.c_void_p(nrbuffer.data_ptr()),
                                                   nrbuffer.stride(0), _stream_ptr(stream)))
        return nrbuffer

    def doConvolutionalProcessing(self, iNRLevel, bFilterEnabled, dFLoCut, dFHiCut, iq, out=None,
                                  stream=None):
        n = iq.shape[1]
        if out is None:
            out = torch.empty((self.n_channels, n // self.decim, 2), dtype=torch.int16,
                              device=iq.device)
        _lib.check(self.lib.rdsp_doConvolutionalProcessing(
            self.h, float(iNRLevel), int(bool(bFilterEnabled)), dFLoCut, dFHiCut,
            C.c_void_p(iq.data_ptr()), iq.stride(0) // 2, n // 128, C.c_void_p(out.data_ptr()),
            out.stride(0) // 2, _stream_ptr(stream)))
        return out

    def set_engine_literal(self, on, tables=None):
        """the reference's own pre-processor and engine in front of this (bare CONV stage) chain: INO:53-54,71-86"""
        _lib.check(self.lib.rdsp_sdr_set_engine_literal(self.h, int(bool(on))))
        if on and tables is not None:
            import numpy as np
            b = np.ascontiguousarray(tables[0], np.float32).reshape(-1)
            h = np.ascontiguousarray(tables[1], np.float32).reshape(-1)
            _lib.check(self.lib.rdsp_sdr_load_engine_tables(self.h, b.ctypes.data_as(_lib._f32p), h.ctypes.data_as(_lib._f32p)))

    def reset(self, stream=None):
        _lib.check(self.lib.rdsp_chain_reset(self.h, _stream_ptr(stream)))

    # ---- AudioSDR-style setters (INO:117-139, CTL:149-423) -------------------
    def enableAGC(self): _lib.check(self.lib.rdsp_sdr_enableAGC(self.h))
    def disableAGC(self): _lib.check(self.lib.rdsp_sdr_disableAGC(self.h))
    def setAGCmode(self, mode): _lib.check(self.lib.rdsp_sdr_setAGCmode(self.h, int(mode)))
    def enableALSfilter(self): _lib.check(self.lib.rdsp_sdr_enableALSfilter(self.h))
    def disableALSfilter(self): _lib.check(self.lib.rdsp_sdr_disableALSfilter(self.h))
    def setALSfilterNotch(self): _lib.check(self.lib.rdsp_sdr_setALSfilterNotch(self.h))
    def setALSfilterPeak(self): _lib.check(self.lib.rdsp_sdr_setALSfilterPeak(self.h))
    def setALSfilterAdaptive(self): _lib.check(self.lib.rdsp_sdr_setALSfilterAdaptive(self.h))
    def enableNoiseBlanker(self): _lib.check(self.lib.rdsp_sdr_enableNoiseBlanker(self.h))
    def disableNoiseBlanker(self): _lib.check(self.lib.rdsp_sdr_disableNoiseBlanker(self.h))
    def setNoiseBlankerThresholdDb(self, db): _lib.check(self.lib.rdsp_sdr_setNoiseBlankerThresholdDb(self.h, float(db)))
    def swapIQ(self, on): _lib.check(self.lib.rdsp_pre_swapIQ(self.h, int(bool(on))))
    def startAutoI2SerrorDetection(self): _lib.check(self.lib.rdsp_pre_startAutoI2SerrorDetection(self.h))
    def setIQslip(self, slip): _lib.check(self.lib.rdsp_pre_setIQslip(self.h, int(slip)))
    def setInputGain(self, g): _lib.check(self.lib.rdsp_sdr_setInputGain(self.h, float(g)))
    def setOutputGain(self, g): _lib.check(self.lib.rdsp_sdr_setOutputGain(self.h, float(g)))
    def setIQgainBalance(self, g): _lib.check(self.lib.rdsp_sdr_setIQgainBalance(self.h, float(g)))
    def enableAudioFilter(self): _lib.check(self.lib.rdsp_sdr_enableAudioFilter(self.h))
    def setAudioFilter(self, f, stream=None):
        _lib.check(self.lib.rdsp_sdr_setAudioFilter(self.h, int(f), _stream_ptr(stream)))
    def setAudioFilterKind(self, kind, stream=None):
        """0: SDR.setAudioFilter() filters are overlap-save masks; 1: 8th-order IIR band-passes (biquads)"""
        _lib.check(self.lib.rdsp_sdr_setAudioFilterKind(self.h, int(kind), _stream_ptr(stream)))

    def setAudioIIRCoefficients(self, coef20, group=None):
        """four sections {b0, b1, b2, a1, a2} (CMSIS order, feedback added) instead of the designed cascade"""
        a = np.ascontiguousarray(coef20, dtype=np.float32).reshape(20)
        if group is None:
            _lib.check(self.lib.rdsp_sdr_setAudioIIRCoefficients(self.h, a.ctypes.data_as(_lib._f32p)))
        else:
            _lib.check(self.lib.rdsp_group_setAudioIIRCoefficients(self.h, int(group), a.ctypes.data_as(_lib._f32p)))

    def iir_coeffs(self, group=0):
        a = np.zeros(20, np.float32)
        _lib.check(self.lib.rdsp_chain_get_iir_coeffs(self.h, int(group), a.ctypes.data_as(_lib._f32p)))
        return a

    def setDemodMode(self, mode, stream=None):
        return int(self.lib.rdsp_sdr_setDemodMode(self.h, int(mode), _stream_ptr(stream)))
    def setMute(self, m): _lib.check(self.lib.rdsp_sdr_setMute(self.h, int(bool(m))))
    def setTuningOffsetHz(self, hz): _lib.check(self.lib.rdsp_sdr_setTuningOffsetHz(self.h, float(hz)))
    def set_nr_level(self, lvl): _lib.check(self.lib.rdsp_set_nr_level(self.h, int(lvl)))
    def set_spectral_nr(self, on, level): _lib.check(self.lib.rdsp_set_spectral_nr(self.h, int(on), float(level)))
    def set_nlms_energy_mode(self, running): _lib.check(self.lib.rdsp_set_nlms_energy_mode(self.h, int(bool(running))))
    def set_spectral_resynthesis(self, literal): _lib.check(self.lib.rdsp_set_spectral_resynthesis(self.h, 2 if literal == 2 else int(bool(literal))))

    # ---- receiver groups: per-group retune / PBT / mode table (CTL:330-423,569-612) --
    def set_groups(self, group_of_channel):
        """Partition the channels into receiver groups (ids 0..G-1); None = one group."""
        if group_of_channel is None:
            _lib.check(self.lib.rdsp_chain_set_groups(self.h, 1, None))
            return
        g = np.ascontiguousarray(group_of_channel, dtype=np.uint16)
        if g.shape != (self.n_channels,):
            raise ValueError("one group id per channel")
        _lib.check(self.lib.rdsp_chain_set_groups(self.h, int(g.max()) + 1, g.ctypes.data_as(C.POINTER(C.c_uint16))))

    @property
    def n_groups(self):
        return int(self.lib.rdsp_chain_groups(self.h))

    def group_reInitializeFilter(self, group, lo, hi, stream=None):
        _lib.check(self.lib.rdsp_group_reInitializeFilter(self.h, int(group), float(lo), float(hi), _stream_ptr(stream)))

    def group_setAudioFilter(self, group, f, stream=None):
        _lib.check(self.lib.rdsp_group_setAudioFilter(self.h, int(group), int(f), _stream_ptr(stream)))

    def group_setDemodMode(self, group, mode, stream=None):
        return int(self.lib.rdsp_group_setDemodMode(self.h, int(group), int(mode), _stream_ptr(stream)))

    def group_setTuningOffsetHz(self, group, hz):
        _lib.check(self.lib.rdsp_group_setTuningOffsetHz(self.h, int(group), float(hz)))

    def group_pbt(self, group, edge, direction, stream=None):
        _lib.check(self.lib.rdsp_group_pbt(self.h, int(group), int(edge), int(direction), _stream_ptr(stream)))

    def group_tuningMode(self, group, mndx, vfo_hz, stream=None):
        return int(self.lib.rdsp_group_tuningMode(self.h, int(group), int(mndx), float(vfo_hz), _stream_ptr(stream)))

    def group_mask(self, group):
        a = np.zeros(2 * self.fft_l, np.float32)
        _lib.check(self.lib.rdsp_group_get_mask(self.h, int(group), a.ctypes.data_as(_lib._f32p)))
        return a

    # ---- pipelined mode: tail of call k overlaps the front of call k+1 ------------
    def set_pipelined(self, on):
        _lib.check(self.lib.rdsp_chain_set_pipelined(self.h, int(bool(on))))

    def set_sub_batch(self, channels):
        """pipelined mode: channels per launch (multiple of 64, 0 = one launch per stage)"""
        _lib.check(self.lib.rdsp_chain_set_sub_batch(self.h, int(channels)))

    def set_front_variant(self, lean):
        _lib.check(self.lib.rdsp_chain_set_front_variant(self.h, int(lean)))

    def set_fir_variant(self, variant):
        """stage A3: 4 frequency domain with one-granule frames (split-invariant bits); -1 (default) that, or 5 for
        calls no tail stage follows; 0 the direct
        form, 2 frequency domain with 448-sample frames (throughput; split-invariant bits for calls that are
        multiples of `granule_blocks`, i.e. whole frames), 5 frequency domain on 16-lane rows (split-invariant;
        for chains without a tail stage)"""
        _lib.check(self.lib.rdsp_chain_set_fir_variant(self.h, int(variant)))

    def set_tail_variant(self, lanes_per_channel, matrix_reduce=None):
        if matrix_reduce is None:
            matrix_reduce = lanes_per_channel == 8
        _lib.check(self.lib.rdsp_chain_set_tail_variant(self.h, int(lanes_per_channel), int(matrix_reduce)))

    def flush(self, stream=None):
        _lib.check(self.lib.rdsp_chain_flush(self.h, _stream_ptr(stream)))

    # ---- per-kernel timing (HIP events on the launch stream) -------------------
    def set_timing(self, on):
        _lib.check(self.lib.rdsp_chain_set_timing(self.h, int(bool(on))))

    def front_kernel_name(self):
        return self.lib.rdsp_chain_front_kernel_name(self.h).decode()

    def get_timing(self):
        f, t, n = C.c_double(), C.c_double(), C.c_int()
        _lib.check(self.lib.rdsp_chain_get_timing(self.h, C.byref(f), C.byref(t), C.byref(n)))
        return f.value, t.value, n.value

    def get_timing_span(self):
        """(ms between the end of the first and of the last timed call, timed calls)"""
        t, n = C.c_double(), C.c_int()
        _lib.check(self.lib.rdsp_chain_get_timing_span(self.h, C.byref(t), C.byref(n)))
        return t.value, n.value

    # ---- per-channel state as data (checkpoint / resume, channels moved between chains) ------
    def save_state(self, first_channel=0, n_channels=None, stream=None):
        """uint8 array holding the DSP state of channels first_channel .. first_channel + n_channels - 1"""
        n = self.n_channels - first_channel if n_channels is None else n_channels
        size = self.lib.rdsp_chain_state_bytes(self.h, n)
        buf = np.zeros(size, np.uint8)
        _lib.check(self.lib.rdsp_chain_save_state(self.h, first_channel, n, buf.ctypes.data_as(C.c_void_p), size,
                                                  _stream_ptr(stream)))
        return buf

    def load_state(self, blob, first_channel=0, stream=None):
        blob = np.ascontiguousarray(blob, np.uint8)
        _lib.check(self.lib.rdsp_chain_load_state(self.h, first_channel, blob.ctypes.data_as(C.c_void_p), blob.size,
                                                  _stream_ptr(stream)))

    # ---- state read-back ------------------------------------------------------
    def scalars(self, stream=None):
        a = np.zeros((self.n_channels, 4), np.float32)
        _lib.check(self.lib.rdsp_chain_get_scalars(self.h, a.ctypes.data_as(_lib._f32p), _stream_ptr(stream)))
        return a

    def lms_coeffs(self, which=0, stream=None):
        a = np.zeros((self.n_channels, 96), np.float32)
        _lib.check(self.lib.rdsp_chain_get_lms_coeffs(self.h, which, a.ctypes.data_as(_lib._f32p), _stream_ptr(stream)))
        return a

    STATUS_NR_ENERGY, STATUS_NR_NONFINITE, STATUS_ALS_ENERGY, STATUS_ALS_NONFINITE = 0x01, 0x02, 0x10, 0x20

    def get_status(self, stream=None):
        """per-channel NLMS health words (include/rdsp.h RDSP_STATUS_*), uint32 [n_channels]"""
        import ctypes as C
        a = np.zeros(self.n_channels, np.uint32)
        _lib.check(self.lib.rdsp_chain_get_status(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), _stream_ptr(stream)))
        return a

    def reset_nlms_channels(self, which, first_channel, n_channels=1, stream=None):
        """boot values for the NLMS instance (0 DSP-NR, 1 ALS) of a range of channels: the cure for a channel
        the health words name (include/rdsp.h rdsp_chain_reset_nlms_channels)"""
        _lib.check(self.lib.rdsp_chain_reset_nlms_channels(self.h, int(which), int(first_channel), int(n_channels), _stream_ptr(stream)))

    def mask(self):
        a = np.zeros(2 * self.fft_l, np.float32)
        _lib.check(self.lib.rdsp_chain_get_mask(self.h, a.ctypes.data_as(_lib._f32p)))
        return a

    def fir_taps(self):
        a = np.zeros(256, np.float32)
        _lib.check(self.lib.rdsp_chain_get_fir_taps(self.h, a.ctypes.data_as(_lib._f32p)))
        return a


def pbt_step(lo, hi, edge, direction):
    """checkPBT_Increase/Decrease (CTL:569-612) on a pair of cut-offs; returns (lo, hi)."""
    lib = _lib.load()
    a, b = C.c_double(float(lo)), C.c_double(float(hi))
    _lib.check(lib.rdsp_pbt_step(C.byref(a), C.byref(b), int(edge), int(direction)))
    return a.value, b.value


def synth_iq(n_channels, n_samples, ch0=0, t0=0, cw=False, n_threads=0, out=None):
    """Deterministic synthetic IQ: int16 numpy [n_channels, n_samples, 2]."""
    lib = _lib.load()
    if out is None:
        out = np.empty((n_channels, n_samples, 2), np.int16)
    sc = synth_config(cw=cw)
    lib.rdsp_synth_iq(out.ctypes.data_as(_lib._i16p), int(ch0), int(n_channels), int(t0),
                      int(n_samples), C.byref(sc), int(n_threads))
    return out


def calc_cplx_FIR_coeffs(numCoeffs, FLoCut, FHiCut, SampleRate, window=1):
    lib = _lib.load()
    ci = np.zeros(numCoeffs)
    cq = np.zeros(numCoeffs)
    lib.rdsp_calc_cplx_FIR_coeffs(ci.ctypes.data_as(_lib._f64p), cq.ctypes.data_as(_lib._f64p),
                                  numCoeffs, FLoCut, FHiCut, SampleRate, window)
    return ci, cq


def init_filter_mask(coef_I, coef_Q, fft_l):
    lib = _lib.load()
    m = np.zeros(2 * fft_l, np.float32)
    rc = lib.rdsp_init_filter_mask(m.ctypes.data_as(_lib._f32p),
                                   np.ascontiguousarray(coef_I).ctypes.data_as(_lib._f64p),
                                   np.ascontiguousarray(coef_Q).ctypes.data_as(_lib._f64p), fft_l)
    _lib.check(rc)
    return m


def estimate_iq_slip(iq):
    """I2S channel-slip estimate of one recorded channel (int16 [n, 2], host): (slip, rejection_db[3]);
    pass `slip` to Chain.setIQslip.  include/rdsp.h rdsp_estimate_iq_slip."""
    import ctypes as C
    lib = _lib.load()
    a = np.ascontiguousarray(iq, dtype=np.int16)
    assert a.ndim == 2 and a.shape[1] == 2
    slip, rej = C.c_int(0), (C.c_double * 3)()
    _lib.check(lib.rdsp_estimate_iq_slip(a.ctypes.data_as(C.POINTER(C.c_int16)), a.shape[0], C.byref(slip), rej))
    return slip.value, [rej[0], rej[1], rej[2]]
