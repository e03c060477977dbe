"""AudioFilterBiquad and AudioAnalyzeFFT1024 of the sketch's graph (RadioDSP_SDR_RX.ino:57-59,
75-78,87,155-156) batched over channels on the GPU; thin ctypes mirrors of include/rdsp.h."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .spectrum import WINDOW_IDS

WINDOWS = {"none": 0, **{f"AudioWindow{k}1024": v for k, v in WINDOW_IDS.items() if v}}


def _stream(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


def biquad_design(kind, freq, q, fs):
    """RBJ section in the cascade's order {b0, b1, b2, -a1, -a2}; kind 0 LP, 1 HP, 2 BP, 3 notch"""
    c = np.zeros(5, np.float32)
    _lib.load().rdsp_biquad_design(int(kind), float(freq), float(q), float(fs), c.ctypes.data_as(_lib._f32p))
    return c


def teensy_biquad_design(kind, frequency, q, fs=44100.0):
    """what AudioFilterBiquad's setLowpass / setHighpass / setBandpass / setNotch compute: int32 x 2^30
    {b0, b1, b2, a1, a2} (setCoefficients negates a1, a2); kind 0 LP, 1 HP, 2 BP, 3 notch"""
    c = np.zeros(5, np.int32)
    _lib.load().rdsp_teensy_biquad_design(int(kind), float(frequency), float(q), float(fs), c.ctypes.data_as(C.POINTER(C.c_int32)))
    return c


def design_audio_iir(f1, f2, fs):
    """the engine's 8th-order band-pass (four sections x {b0, b1, b2, -a1, -a2})"""
    c = np.zeros(20, np.float32)
    _lib.load().rdsp_design_audio_iir(float(f1), float(f2), float(fs), c.ctypes.data_as(_lib._f32p))
    return c


class FilterBiquad:
    """AudioFilterBiquad of the Teensy Audio library: up to four cascaded fixed-point sections (coefficients x 2^30,
    32 x 16 products, 14-bit error feedback), int16 audio in and out.  A fresh object passes nothing; update() runs
    stage 0 and every further stage that was chained on by a setter call for it."""

    def __init__(self, n_channels, fs=44100.0, device=0):
        self.lib = _lib.load()
        self.n_channels = int(n_channels)
        h = C.c_void_p()
        _lib.check(self.lib.rdsp_biquad_create(self.n_channels, int(device), float(fs), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.rdsp_biquad_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def setCoefficients(self, stage, coefficients):
        """five doubles {b0, b1, b2, a1, a2} of (b0 + b1/z + b2/z^2) / (1 + a1/z + a2/z^2), or five ints already x 2^30"""
        c = np.asarray(coefficients)
        assert c.shape == (5,)
        if np.issubdtype(c.dtype, np.integer):
            ci = np.ascontiguousarray(c, np.int32)
            _lib.check(self.lib.rdsp_biquad_setCoefficients_int(self.h, int(stage), ci.ctypes.data_as(C.POINTER(C.c_int32))))
        else:
            cd = np.ascontiguousarray(c, np.float64)
            _lib.check(self.lib.rdsp_biquad_setCoefficients(self.h, int(stage), cd.ctypes.data_as(_lib._f64p)))

    def definition(self):
        """(int32 [4, 5] coefficient words as the library's definition[] holds them, stages the cascade runs)"""
        a, n = np.zeros(20, np.int32), C.c_int()
        _lib.check(self.lib.rdsp_biquad_get_definition(self.h, a.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)))
        return a.reshape(4, 5), n.value

    def setLowpass(self, stage, frequency, q=0.7071):
        _lib.check(self.lib.rdsp_biquad_setLowpass(self.h, int(stage), float(frequency), float(q)))

    def setHighpass(self, stage, frequency, q=0.7071):
        _lib.check(self.lib.rdsp_biquad_setHighpass(self.h, int(stage), float(frequency), float(q)))

    def setBandpass(self, stage, frequency, q=1.0):
        _lib.check(self.lib.rdsp_biquad_setBandpass(self.h, int(stage), float(frequency), float(q)))

    def setNotch(self, stage, frequency, q=1.0):
        _lib.check(self.lib.rdsp_biquad_setNotch(self.h, int(stage), float(frequency), float(q)))

    def coeffs(self):
        c = np.zeros(20, np.float32)
        _lib.check(self.lib.rdsp_biquad_get_coeffs(self.h, c.ctypes.data_as(_lib._f32p)))
        return c

    def update(self, audio, stream=None):
        """audio: int16 cuda tensor [n_channels, n] (mono) or [n_channels, n, 2] (each side filtered
        as its own pass is up to the caller: pass audio[..., 0] views); returns the filtered int16."""
        assert audio.is_cuda and audio.dtype == torch.int16 and audio.shape[0] == self.n_channels
        assert audio.dim() == 2 and audio.stride(1) in (1, 2) and audio.shape[1] % 128 == 0
        step = audio.stride(1)
        out = torch.empty((self.n_channels, audio.shape[1]), dtype=torch.int16, device=audio.device)
        _lib.check(self.lib.rdsp_biquad_update(self.h, C.c_void_p(audio.data_ptr()), audio.stride(0) // step, step,
                                               audio.shape[1] // 128, C.c_void_p(out.data_ptr()), out.stride(0), 1,
                                               _stream(stream)))
        return out


class AnalyzeFFT1024:
    """AudioAnalyzeFFT1024: 1024-point frames (hop 512) of an int16 audio stream, 512 magnitudes."""

    def __init__(self, n_channels, window="AudioWindowHanning1024", device=0):
        self.lib = _lib.load()
        self.n_channels = int(n_channels)
        h = C.c_void_p()
        _lib.check(self.lib.rdsp_fft1024_create(self.n_channels, int(device), WINDOWS[window], C.byref(h)))
        self.h = h
        self.output = None
        self._flag = False

    def close(self):
        if getattr(self, "h", None):
            self.lib.rdsp_fft1024_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def windowFunction(self, window):
        """A table name, an int16 array of 1024 q15 taps (the library's own argument) or None."""
        if window is None or isinstance(window, str):
            _lib.check(self.lib.rdsp_fft1024_windowFunction(self.h, WINDOWS[window or "none"]))
            return
        w = np.ascontiguousarray(window, dtype=np.int16)
        assert w.shape == (1024,)
        _lib.check(self.lib.rdsp_fft1024_windowFunction_table(self.h, w.ctypes.data_as(C.POINTER(C.c_int16))))

    def averageTogether(self, n):   # INO:148; the library ignores it
        _lib.check(self.lib.rdsp_fft1024_averageTogether(self.h, int(n)))

    def update(self, audio, stream=None):
        """audio: int16 cuda tensor [n_channels, n_blocks*128], sample stride 1 or 2 (e.g. the L side
        out[..., 0] of the chain's interleaved output).  Returns int16-storage [n_channels, n_out, 512]
        (view as uint16 on the host)."""
        assert audio.is_cuda and audio.dtype == torch.int16 and audio.dim() == 2 and audio.shape[0] == self.n_channels
        step = audio.stride(1)
        assert step in (1, 2) and audio.shape[1] % 128 == 0
        nb = audio.shape[1] // 128
        n_out = self.lib.rdsp_fft1024_outputs_for(self.h, nb)
        out = torch.zeros((self.n_channels, max(n_out, 1), 512), dtype=torch.int16, device=audio.device)
        got = C.c_int()
        _lib.check(self.lib.rdsp_fft1024_update(self.h, C.c_void_p(audio.data_ptr()), audio.stride(0) // step, step, nb,
                                                C.c_void_p(out.data_ptr()), out.shape[1], C.byref(got), _stream(stream)))
        out = out[:, :got.value]
        if got.value:
            self.output = out[:, -1]
            self._flag = True
        return out

    def available(self):
        f, self._flag = self._flag, False
        return f

    def read(self, channel, binFirst, binLast=None):
        if self.output is None:
            return 0.0
        row = np.ascontiguousarray(self.output[channel].cpu().numpy().view(np.uint16)).ctypes.data_as(C.POINTER(C.c_uint16))
        if binLast is None:
            return float(self.lib.rdsp_fft1024_read(row, int(binFirst)))
        return float(self.lib.rdsp_fft1024_read_range(row, int(binFirst), int(binLast)))
