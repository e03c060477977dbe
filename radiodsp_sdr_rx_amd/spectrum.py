"""AudioAnalyzeFFT256IQ (analyze_fft256iq.h:52-110) batched over channels: the IQ
panadapter spectrum analyser on the GPU (integer q15 path, bit-exact against the CPU restatement kept with the tests)."""
import ctypes as C

import torch

from . import _lib

WINDOWS = {"none": 0, "AudioWindowHanning256": 1, "AudioWindowBlackmanHarris256": 2}


class AnalyzeFFT256IQ:
    def __init__(self, n_channels, naverage=8, window="AudioWindowHanning256", device=0):
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
        _lib.check(self.lib.rdsp_spectrum_windowFunction(self.h, WINDOWS[window]))

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

    def read(self, channel, binNumber):  # FFTIQ.h:70-73
        if binNumber > 255 or self.output is None:
            return 0.0
        return float(int(self.output[channel, binNumber].item()) & 0xFFFF) * (1.0 / 16384.0)
