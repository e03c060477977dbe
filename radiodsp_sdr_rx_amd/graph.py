"""Host-side mirror of the reference's block graph (AudioStream nodes, AudioConnection,
AudioMemory, record/play queues; RadioDSP_SDR_RX.ino:52-89,151) over the C runtime in
csrc/rdsp_graph.c.  A block is an int16 tile [n_channels][128]."""
import ctypes as C

import numpy as np

from . import _lib

BLOCK = 128


class Graph:
    def __init__(self, n_channels=1):
        self.lib = _lib.load()
        self.n_channels = n_channels
        self.h = C.c_void_p(self.lib.rdsp_graph_create(n_channels))
        assert self.h
        self._keep = []  # callbacks / chains must outlive the graph

    def close(self):
        if self.h:
            self.lib.rdsp_graph_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def AudioMemory(self, n):
        _lib.check(self.lib.rdsp_memory(self.h, n))

    def memory_usage(self):
        return self.lib.rdsp_memory_usage(self.h), self.lib.rdsp_memory_usage_max(self.h)

    def node(self, ninputs, update):
        """AudioStream subclass: update(node) is called once per tick."""
        n = Node(self, None)

        def tramp(node_ptr, _user):
            update(n)

        cb = _lib.UPDATE_FN(tramp)
        self._keep.append(cb)
        n.h = C.c_void_p(self.lib.rdsp_node_create(self.h, ninputs, cb, None))
        assert n.h
        return n

    def input_node(self):
        return InputNode(self, C.c_void_p(self.lib.rdsp_input_node_create(self.h)))

    def record_queue(self):
        return RecordQueue(self, C.c_void_p(self.lib.rdsp_record_queue_create(self.h)))

    def play_queue(self):
        return PlayQueue(self, C.c_void_p(self.lib.rdsp_play_queue_create(self.h)))

    def sdr_node(self, chain):
        h = self.lib.rdsp_sdr_node_create(self.h, chain.h)
        if not h:
            raise _lib.RdspError(-1, self.lib.rdsp_last_error().decode())
        self._keep.append(chain)
        return Node(self, C.c_void_p(h))

    def engine_node(self, engine):
        """`AudioSDR SDR;` as the reference's engine computes it (INO:54, wired INO:81-86): radiodsp_sdr_rx_amd.engine.Engine"""
        h = self.lib.rdsp_engine_node_create(self.h, engine.h)
        if not h:
            raise _lib.RdspError(-1, self.lib.rdsp_last_error().decode())
        self._keep.append(engine)
        return EngineNode(self, C.c_void_p(h))

    def preproc_node(self, pre):
        """`AudioSDRpreProcessor preProcessor;` (INO:53, wired INO:71-72): radiodsp_sdr_rx_amd.engine.PreProcessor"""
        h = self.lib.rdsp_preproc_node_create(self.h, pre.h)
        if not h:
            raise _lib.RdspError(-1, self.lib.rdsp_last_error().decode())
        self._keep.append(pre)
        return EngineNode(self, C.c_void_p(h))

    def spectrum_node(self, analyser):
        """AudioAnalyzeFFT256IQ as a node (INO:57,73-74): inputs I, Q; available()/output like FFTIQ.h"""
        h = self.lib.rdsp_spectrum_node_create(self.h, analyser.h)
        if not h:
            raise _lib.RdspError(-1, self.lib.rdsp_last_error().decode())
        self._keep.append(analyser)
        return SpectrumNode(self, C.c_void_p(h))

    def biquad_node(self, biquad):
        """AudioFilterBiquad as a node (INO:58-59,75-78): one input, one output"""
        h = self.lib.rdsp_biquad_node_create(self.h, biquad.h)
        if not h:
            raise _lib.RdspError(-1, self.lib.rdsp_last_error().decode())
        self._keep.append(biquad)
        return BiquadNode(self, C.c_void_p(h))

    def fft1024_node(self, analyser):
        """AudioAnalyzeFFT1024 as a node (INO:57,87): one input; available()/output like the library"""
        h = self.lib.rdsp_fft1024_node_create(self.h, analyser.h)
        if not h:
            raise _lib.RdspError(-1, self.lib.rdsp_last_error().decode())
        self._keep.append(analyser)
        return Fft1024Node(self, C.c_void_p(h))

    def AudioConnection(self, src, src_port, dst, dst_port):
        _lib.check(self.lib.rdsp_connect(src.h, src_port, dst.h, dst_port))

    def update_all(self):
        return self.lib.rdsp_update_all(self.h)

    def AudioNoInterrupts(self):
        self.lib.rdsp_no_interrupts(self.h)

    def AudioInterrupts(self):
        self.lib.rdsp_interrupts(self.h)


class Block:
    def __init__(self, graph, h):
        self.g, self.h = graph, C.c_void_p(h)

    def data(self):
        p = self.g.lib.rdsp_block_data(self.h)
        return np.ctypeslib.as_array(p, (self.g.n_channels, BLOCK))

    def refcount(self):
        return self.g.lib.rdsp_block_refcount(self.h)


class Node:
    def __init__(self, graph, h):
        self.g, self.h = graph, h

    def allocate(self):
        h = self.g.lib.rdsp_allocate(self.h)
        return Block(self.g, h) if h else None

    def receiveReadOnly(self, port):
        h = self.g.lib.rdsp_receive_readonly(self.h, port)
        return Block(self.g, h) if h else None

    def receiveWritable(self, port):
        h = self.g.lib.rdsp_receive_writable(self.h, port)
        return Block(self.g, h) if h else None

    def transmit(self, block, port=0):
        self.g.lib.rdsp_transmit(self.h, block.h, port)

    def release(self, block):
        if block is not None:
            self.g.lib.rdsp_release(block.h)

    def status(self):
        return self.g.lib.rdsp_sdr_node_status(self.h)


class EngineNode(Node):
    def status(self):
        return self.g.lib.rdsp_engine_node_status(self.h)


class SpectrumNode(Node):
    def available(self):  # FFTIQ.h:62-68
        return bool(self.g.lib.rdsp_spectrum_node_available(self.h))

    def output(self):     # FFTIQ.h:99, uint16 [n_channels, 256]
        p = self.g.lib.rdsp_spectrum_node_output(self.h)
        return np.ctypeslib.as_array(p, (self.g.n_channels, 256)).copy()

    def read(self, channel, binFirst, binLast=None):  # FFTIQ.h:70-73 and :75-86 (binLast itself is not added)
        if binLast is None:
            return float(self.g.lib.rdsp_spectrum_node_read(self.h, int(channel), int(binFirst)))
        return float(self.g.lib.rdsp_spectrum_node_read_range(self.h, int(channel), int(binFirst), int(binLast)))

    def status(self):
        return self.g.lib.rdsp_spectrum_node_status(self.h)


class BiquadNode(Node):
    def status(self):
        return self.g.lib.rdsp_biquad_node_status(self.h)


class Fft1024Node(Node):
    def available(self):
        return bool(self.g.lib.rdsp_fft1024_node_available(self.h))

    def output(self):     # uint16 [n_channels, 512]
        p = self.g.lib.rdsp_fft1024_node_output(self.h)
        return np.ctypeslib.as_array(p, (self.g.n_channels, 512)).copy()

    def read(self, channel, binFirst, binLast=None):  # AudioAnalyzeFFT1024::read (the range form includes binLast)
        if binLast is None:
            return float(self.g.lib.rdsp_fft1024_node_read(self.h, int(channel), int(binFirst)))
        return float(self.g.lib.rdsp_fft1024_node_read_range(self.h, int(channel), int(binFirst), int(binLast)))

    def status(self):
        return self.g.lib.rdsp_fft1024_node_status(self.h)


class InputNode(Node):
    def push(self, i_tile, q_tile):
        self._i = np.ascontiguousarray(i_tile, np.int16)
        self._q = np.ascontiguousarray(q_tile, np.int16)
        _lib.check(self.g.lib.rdsp_input_node_push(self.h, self._i.ctypes.data_as(_lib._i16p),
                                                   self._q.ctypes.data_as(_lib._i16p)))


class RecordQueue(Node):
    def begin(self):
        self.g.lib.rdsp_record_queue_begin(self.h)

    def end(self):
        self.g.lib.rdsp_record_queue_end(self.h)

    def available(self):
        return self.g.lib.rdsp_record_queue_available(self.h)

    def readBuffer(self):
        p = self.g.lib.rdsp_record_queue_readBuffer(self.h)
        return np.ctypeslib.as_array(p, (self.g.n_channels, BLOCK)) if p else None

    def freeBuffer(self):
        self.g.lib.rdsp_record_queue_freeBuffer(self.h)


class PlayQueue(Node):
    def getBuffer(self):
        p = self.g.lib.rdsp_play_queue_getBuffer(self.h)
        return np.ctypeslib.as_array(p, (self.g.n_channels, BLOCK)) if p else None

    def playBuffer(self):
        return self.g.lib.rdsp_play_queue_playBuffer(self.h)
