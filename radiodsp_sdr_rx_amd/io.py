"""Recorded IQ in, audio out: thin wrappers over the C-ABI's
rdsp_iq_reader_* / rdsp_audio_writer_* / rdsp_stream_run_* (include/rdsp.h) -- the
I2S/codec ends of the sketch's graph (RadioDSP_SDR_RX.ino:52,55,159-169) for a host
that has files.  All the work is in the library; there is no Python data path."""
import ctypes as C

import numpy as np

from . import _lib

IO_AUTO, IO_RAW, IO_WAV = 0, 1, 2


class IqReader:
    """One receiver channel's recording: RAW int16 I,Q pairs or 16-bit stereo WAV (I left, Q right)."""

    def __init__(self, path, fmt=IO_AUTO):
        self.h = None
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.rdsp_iq_reader_open(str(path).encode(), int(fmt), C.byref(h)))
        self.h = h

    @property
    def sample_rate(self): return float(self.lib.rdsp_iq_reader_sample_rate(self.h))
    @property
    def frames(self): return int(self.lib.rdsp_iq_reader_frames(self.h))
    @property
    def format(self): return int(self.lib.rdsp_iq_reader_format(self.h))

    def read(self, n_pairs):
        buf = np.empty((n_pairs, 2), np.int16)
        got = self.lib.rdsp_iq_reader_read(self.h, buf.ctypes.data_as(_lib._i16p), n_pairs)
        return buf[:got]

    def close(self):
        if self.h:
            self.lib.rdsp_iq_reader_close(self.h)
            self.h = None

    __del__ = close


class AudioWriter:
    """int16 L,R pairs to a RAW or WAV file at the decimated rate."""

    def __init__(self, path, fmt=IO_WAV, sample_rate=24000.0):
        self.h = None
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.rdsp_audio_writer_open(str(path).encode(), int(fmt), float(sample_rate), C.byref(h)))
        self.h = h

    def write(self, lr):
        lr = np.ascontiguousarray(lr, dtype=np.int16)
        return int(self.lib.rdsp_audio_writer_write(self.h, lr.ctypes.data_as(_lib._i16p), lr.shape[0]))

    @property
    def frames(self): return int(self.lib.rdsp_audio_writer_frames(self.h))

    def close(self):
        if self.h:
            h, self.h = self.h, None
            _lib.check(self.lib.rdsp_audio_writer_close(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _stats(st):
    return {k: getattr(st, k) for k, _ in st._fields_}


def stream_files(chain, readers, writers, blocks_per_call, max_blocks=0):
    """rdsp_stream_run_files: every channel from its reader to its writer."""
    lib = _lib.load()
    n = chain.n_channels
    assert len(readers) == n and len(writers) == n
    ra = (C.c_void_p * n)(*[r.h for r in readers])
    wa = (C.c_void_p * n)(*[w.h for w in writers])
    st = _lib.StreamStats()
    _lib.check(lib.rdsp_stream_run_files(chain.h, ra, wa, int(blocks_per_call), int(max_blocks), C.byref(st)))
    return _stats(st)


def stream_memory(chain, iq, blocks_per_call, out=None):
    """rdsp_stream_run_memory: host int16 [n_channels, n_samples, 2] -> host int16 [n_channels, n_samples/decim, 2].
    numpy arrays go through pinned staging slots; pinned torch tensors (pin_memory=True) at both
    ends are used by the DMA engines directly."""
    lib = _lib.load()
    is_torch = hasattr(iq, "data_ptr")
    if not is_torch:
        iq = np.ascontiguousarray(iq, dtype=np.int16)
    nch, n, _ = iq.shape
    assert nch == chain.n_channels and n % 128 == 0
    decim = int(lib.rdsp_chain_decim(chain.h))
    if out is None:
        if is_torch:
            import torch
            out = torch.zeros((nch, n // decim, 2), dtype=torch.int16, pin_memory=iq.is_pinned())
        else:
            out = np.zeros((nch, n // decim, 2), np.int16)
    ptr = (lambda a: C.cast(a.data_ptr(), _lib._i16p)) if is_torch else (lambda a: a.ctypes.data_as(_lib._i16p))
    st = _lib.StreamStats()
    _lib.check(lib.rdsp_stream_run_memory(chain.h, ptr(iq), n, n // 128, ptr(out), out.shape[1], int(blocks_per_call),
                                          C.byref(st)))
    return out, _stats(st)


def stream_callbacks(chain, source, sink, blocks_per_call, max_blocks=0):
    """rdsp_stream_run with Python callables (tests): source(dst[nch, n, 2]) -> blocks delivered,
    sink(src[nch, n_pairs, 2])."""
    lib = _lib.load()
    nch = chain.n_channels

    def _src(_user, dst, stride, n_blocks):
        a = np.ctypeslib.as_array(dst, (nch, stride, 2))
        return int(source(a[:, :n_blocks * 128]))

    def _snk(_user, src, stride, n_pairs):
        a = np.ctypeslib.as_array(src, (nch, stride, 2))
        sink(a[:, :n_pairs].copy())
        return n_pairs

    st = _lib.StreamStats()
    cs, ck = _lib.SOURCE_FN(_src), _lib.SINK_FN(_snk)
    _lib.check(lib.rdsp_stream_run(chain.h, cs, None, ck, None, int(blocks_per_call), int(max_blocks), C.byref(st)))
    return _stats(st)
