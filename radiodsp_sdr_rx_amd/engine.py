"""`AudioSDR SDR;` of the reference sketch (RadioDSP_SDR_RX.ino:54) for many receivers on one GPU: a ctypes mirror of
the rdsp_engine_* entry points of include/rdsp.h.  Method names, argument meaning and numbering are the engine's
(INO:117-139, RDSP_controls.h:149-423); the arithmetic is csrc/rdsp_engine.hip's."""
import ctypes as C

import numpy as np

from ._lib import check, load

LSBmode, USBmode, CW_LSBmode, CW_USBmode, AMmode, SAMmode = range(6)      # as the compiled tuningMode() passes them
audioAM, audioCW, audio2100, audio2700, audio3100, audioNone = 0, 1, 3, 6, 8, 10
AGCoff, AGCfast, AGCmedium, AGCslow = range(4)
_F32P = C.POINTER(C.c_float)


class Engine:
    def __init__(self, n_channels, max_blocks_per_call=64, device=0, tables=None):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.rdsp_engine_create(n_channels, device, max_blocks_per_call, C.byref(h)))
        self.h, self.n_channels, self.max_blocks = h, n_channels, max_blocks_per_call
        if tables is not None:
            self.load_tables(*tables)

    def close(self):
        if self.h:
            self.lib.rdsp_engine_destroy(self.h)
            self.h = None

    __del__ = close

    def load_tables(self, biquad_sets, hilbert64):
        b = np.ascontiguousarray(biquad_sets, np.float32).reshape(-1)
        h = np.ascontiguousarray(hilbert64, np.float32).reshape(-1)
        assert b.size == 300 and h.size == 64
        check(self.lib.rdsp_engine_load_tables(self.h, b.ctypes.data_as(_F32P), h.ctypes.data_as(_F32P)))

    def sketch_setup(self):
        """INO:120-139 in the sketch's order"""
        self.enableAGC(); self.setAGCmode(AGCmedium); self.disableALSfilter(); self.disableNoiseBlanker()
        self.setInputGain(1.0); self.setOutputGain(0.5); self.setIQgainBalance(1.020)
        self.enableAudioFilter(); self.setAudioFilter(audio2700)
        return self.setDemodMode(LSBmode)

    def setDemodMode(self, mode):
        return float(self.lib.rdsp_engine_setDemodMode(self.h, int(mode)))

    def update(self, d_iq, out=None, stream=None):
        """d_iq: torch int16 [n_channels, n, 2] on the engine's device, n a multiple of 128 -> int16 [n_channels, n, 2]"""
        import torch
        nch, n, two = d_iq.shape
        assert nch == self.n_channels and two == 2 and n % 128 == 0 and d_iq.dtype == torch.int16 and d_iq.is_contiguous()
        if out is None:
            out = torch.empty_like(d_iq)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.rdsp_engine_update(self.h, d_iq.data_ptr(), n, n // 128, out.data_ptr(), n, C.c_void_p(s)))
        return out

    def set_groups(self, first_channels):
        """groups of consecutive channels with settings of their own: first_channels[g] = group g's first channel"""
        a = (C.c_int * len(first_channels))(*[int(x) for x in first_channels])
        check(self.lib.rdsp_engine_set_groups(self.h, len(first_channels), a))

    def select_group(self, group):
        """the group the setters address from now on; -1: all"""
        check(self.lib.rdsp_engine_select_group(self.h, int(group)))

    def save_state(self, first_channel, n_channels):
        n = self.lib.rdsp_engine_state_bytes(self.h, n_channels)
        buf = np.zeros(n, np.uint8)
        check(self.lib.rdsp_engine_save_state(self.h, first_channel, n_channels, buf.ctypes.data_as(C.c_void_p), n, None))
        return buf

    def load_state(self, first_channel, blob):
        blob = np.ascontiguousarray(blob, np.uint8)
        check(self.lib.rdsp_engine_load_state(self.h, first_channel, blob.ctypes.data_as(C.c_void_p), blob.size, None))

    def scalars(self):
        o = np.zeros((self.n_channels, 8), np.float32)
        check(self.lib.rdsp_engine_get_scalars(self.h, o.ctypes.data_as(_F32P), None))
        return o

    def agc_curve(self):
        return np.ctypeslib.as_array(self.lib.rdsp_engine_agc_curve(self.h), (130,)).copy()

    def sine_table(self):
        return np.ctypeslib.as_array(self.lib.rdsp_engine_sine_table(self.h), (257,)).copy()

    def reset(self):
        check(self.lib.rdsp_engine_reset(self.h, None))


def _setter(name):
    def f(self, *a):
        check(getattr(self.lib, "rdsp_engine_" + name)(self.h, *a))
    f.__name__ = name
    return f


for _n in ("enableAGC", "setAGCmode", "enableALSfilter", "disableALSfilter", "setALSfilterNotch", "setALSfilterPeak",
           "setALSfilterAdaptive", "enableNoiseBlanker", "disableNoiseBlanker", "setInputGain", "setOutputGain",
           "setIQgainBalance", "enableAudioFilter", "setAudioFilter", "setMute"):
    setattr(Engine, _n, _setter(_n))


class PreProcessor:
    """`AudioSDRpreProcessor preProcessor;` (INO:53): rdsp_preproc_* of include/rdsp.h"""

    def __init__(self, n_channels, device=0):
        self.lib = load()
        h = C.c_void_p()
        check(self.lib.rdsp_preproc_create(n_channels, device, C.byref(h)))
        self.h, self.n_channels = h, n_channels

    def close(self):
        if self.h:
            self.lib.rdsp_preproc_destroy(self.h)
            self.h = None

    __del__ = close

    def startAutoI2SerrorDetection(self):
        check(self.lib.rdsp_preproc_startAutoI2SerrorDetection(self.h))

    def swapIQ(self, on):
        check(self.lib.rdsp_preproc_swapIQ(self.h, int(bool(on))))

    def update(self, d_iq, out=None, stream=None):
        import torch
        nch, n, two = d_iq.shape
        assert nch == self.n_channels and two == 2 and n % 128 == 0 and d_iq.dtype == torch.int16 and d_iq.is_contiguous()
        if out is None:
            out = torch.empty_like(d_iq)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        check(self.lib.rdsp_preproc_update(self.h, d_iq.data_ptr(), n, n // 128, out.data_ptr(), n, C.c_void_p(s)))
        return out

    def state(self):
        o = np.zeros((self.n_channels, 4), np.int16)
        check(self.lib.rdsp_preproc_get_state(self.h, o.ctypes.data_as(C.POINTER(C.c_int16)), None))
        return o
