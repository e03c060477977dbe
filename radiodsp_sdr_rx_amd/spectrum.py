"""AudioAnalyzeFFT256IQ (analyze_fft256iq.h:52-110) batched over channels: the IQ
panadapter spectrum analyser on the GPU (integer q15 path, bit-exact against the CPU restatement kept with the tests)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

# the table names of analyze_fft256iq.h:30-50 -> RDSP_WINDOW_* (include/rdsp.h)
WINDOW_IDS = {"none": 0, "Hanning": 1, "BlackmanHarris": 2, "BlackmanNuttall": 3, "Bartlett": 4, "Blackman": 5,
              "Flattop": 6, "Nuttall": 7, "Welch": 8, "Hamming": 9, "Cosine": 10, "Tukey": 11}
WINDOWS = {"none": 0, **{f"AudioWindow{k}256": v for k, v in WINDOW_IDS.items() if v}}


def window_q15(window_id, n=256):
    """The q15 table the Teensy Audio library holds under that name (int16 [n])."""
    w = np.zeros(n, np.int16)
    _lib.load().rdsp_window_q15_n(int(window_id), int(n), w.ctypes.data_as(C.POINTER(C.c_int16)))
    return w


class AnalyzeFFT256IQ:
    """`AnalyzeFFT256IQ(n)` is the reference's constructor (FFTIQ.h:55-58): BlackmanNuttall256, naverage 8."""

    def __init__(self, n_channels, naverage=8, window="AudioWindowBlackmanNuttall256", device=0):
        self.lib = _lib.load()
        self.n_channels = n_channels
        h = C.c_void_p()
        _lib.check(self.lib.rdsp_spectrum_create(n_channels, device, naverage, WINDOWS[window], C.byref(h)))
        self.h = h
        self.output = None      # uint16 [n_channels, 256] of the latest completed spectrum
        self._flag = False

    def close(self):
        if getattr(self, "h", None):
            self.lib.rdsp_spectrum_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def averageTogether(self, n):
        _lib.check(self.lib.rdsp_spectrum_averageTogether(self.h, int(n)))

    def windowFunction(self, window):
        """FFTIQ.h:93-95: a table name, an int16 array of 256 q15 taps (the reference's own argument) or None."""
        if window is None or isinstance(window, str):
            _lib.check(self.lib.rdsp_spectrum_windowFunction(self.h, WINDOWS[window or "none"]))
            return
        w = np.ascontiguousarray(window, dtype=np.int16)
        assert w.shape == (256,)
        _lib.check(self.lib.rdsp_spectrum_windowFunction_table(self.h, w.ctypes.data_as(C.POINTER(C.c_int16))))

    def update(self, iq, stream=None):
        """iq: int16 cuda tensor [n_channels, n_blocks*128, 2]; returns uint16-valued int16-storage
        tensor [n_channels, n_out, 256] (viewed as uint16 via .cpu().numpy().view('uint16'))."""
        assert iq.is_cuda and iq.dtype == torch.int16 and iq.is_contiguous() and iq.shape[0] == self.n_channels
        nb = iq.shape[1] // 128
        n_out = self.lib.rdsp_spectrum_outputs_for(self.h, nb)
        out = torch.zeros((self.n_channels, max(n_out, 1), 256), dtype=torch.int16, device=iq.device)
        got = C.c_int()
        s = stream if stream is not None else torch.cuda.current_stream()
        _lib.check(self.lib.rdsp_spectrum_update(self.h, C.c_void_p(iq.data_ptr()), iq.stride(0) // 2, nb,
                                                 C.c_void_p(out.data_ptr()), out.shape[1], C.byref(got),
                                                 C.c_void_p(s.cuda_stream)))
        out = out[:, :got.value]
        if got.value:
            self.output = out[:, -1]
            self._flag = True
        return out

    def available(self):  # FFTIQ.h:62-68
        f, self._flag = self._flag, False
        return f

    def _row(self, channel):
        return np.ascontiguousarray(self.output[channel].cpu().numpy().view(np.uint16))

    def read(self, channel, binFirst, binLast=None):  # FFTIQ.h:70-73 and :75-86
        if self.output is None:
            return 0.0
        row = self._row(channel).ctypes.data_as(C.POINTER(C.c_uint16))
        if binLast is None:
            return float(self.lib.rdsp_spectrum_read(row, int(binFirst)))
        return float(self.lib.rdsp_spectrum_read_range(row, int(binFirst), int(binLast)))
