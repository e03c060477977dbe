"""The reference's AudioSDR engine and AudioSDRpreProcessor, run out of the firmware image (build container only).

`SDR` and `preProcessor` of the sketch (INO:53-54) are objects of Derek Rowell's AudioSDR library, which is not in the
reference tree; their compiled methods are in pre_compiled/RadioDSP_SDR_RX.ino.hex and run under tests/golden/thumb_emu.py
like the rest of the image.  This module sets the objects up the way the sketch does (INO:117-139), feeds them int16 blocks
through hooks that stand in for AudioStream's receive / transmit / release, and can record the float buffers between the
stages of AudioSDR::update() (ITCM 0xe730) -- the stage taps the restatement in oracle/rdsp_engine_oracle.c is written
against.  Used by make_engine_kat.py and make_engine_blackbox.py; nothing of the image is stored, only what it computes.

ITCM addresses (this image only; its SHA-256 is checked by make_firmware_kat.py):
  AudioSDR::AudioSDR 0x6744 -> init 0xede4 (-> AGC init 0xdf14 -> gain curve 0xdd40; SAM constants 0xed34)
  ::update 0xe730, frequency shifter 0xd600, noise blanker 0xe14c, audio filter 0xd944, AGC 0xdb58, ALS filter 0xda24,
  synchronous AM 0xe390;  setters as in METHODS below
  AudioSDRpreProcessor::update 0xee88, ::startAutoI2SerrorDetection 0xf084
"""
import numpy as np

import make_firmware_kat as M
from firmware_image import Image

SDR = 0x20017208                       # the sketch's `SDR` object (.bss)
PRE = 0x20016d50                       # scratch home for a pre-processor object of our own (its constructor is run on it)
METHODS = dict(ctor=0x6744, init=0xede4, update=0xe730, enableAGC=0xdfd4, setAGCmode=0xdfe0, enableALSfilter=0xdb2c,
               disableALSfilter=0xdb14, setALSfilterNotch=0xdb1c, setALSfilterAdaptive=0xdb24, disableNoiseBlanker=0xe380,
               setInputGain=0xd8a0, setOutputGain=0xd918, setIQgainBalance=0xd8f0, enableAudioFilter=0xd970,
               setAudioFilter=0xd97c, setDemodMode=0xd798, setMute=0xd924, allocate=0x10cd4)
FLOAT_ARG = ("setInputGain", "setOutputGain", "setIQgainBalance")
# where update() stands when a stage has just finished, and which float buffers of the object to copy there
TAPS = {0xe7f2: ("in", ("I", "Q")), 0xe804: ("nb", ("I", "Q")), 0xe824: ("pre", ("I", "Q")), 0xea7e: ("mix", ("I", "Q")),
        0xeb9a: ("hilbert", ("I", "Q")), 0xe84e: ("demod", ("A",)), 0xe860: ("filt", ("A",)), 0xe872: ("agc", ("A",)),
        0xe882: ("als", ("A",))}
BUF = dict(A=0x30, I=0x230, Q=0x430)
# the engine's state that lives outside the object (statics of the library's translation unit)
GLOBALS = dict(nco_phase=0x200213b0, am_phase=0x200213ac, i_line=0x200203a0, q_line=0x20020ba4, sam_cos=0x200213a4,
               sam_sin=0x20020ba0, sam_u=0x200213a8, sam_err=0x200213b4)


class EngineRef:
    def __init__(self, im=None, sketch_setup=True, taps=False):
        self.r = r = M.Ref(im or Image())
        self.bl = M.Blocks(r, 64)
        self.cur, self.sent = {}, {}
        h = r.cpu.hooks
        h[M.A["receiveReadOnly"]] = lambda c: c.r.__setitem__(0, self.cur[c.r[1]])
        h[M.A["receiveWritable"]] = lambda c: c.r.__setitem__(0, self.cur[c.r[1]])
        h[M.A["release"]] = lambda c: None
        h[METHODS["allocate"]] = lambda c: c.r.__setitem__(0, self.bl.new())
        h[M.A["transmit"]] = lambda c: self.sent.__setitem__(c.r[2], r.get(c.r[1] + 4, 128, np.int16))
        r.call_addr(METHODS["ctor"], SDR)
        self.tuning_offset = None
        self.taps = {} if taps else None
        if taps:
            for addr, (name, bufs) in TAPS.items():
                r.cpu.watch[addr] = (lambda c, name=name, bufs=bufs: self.taps.setdefault(name, []).append(
                    np.stack([r.get(SDR + BUF[b], 128, np.float32) for b in bufs])))
        if sketch_setup:                                        # INO:120-139 in the sketch's order
            self.call("enableAGC")
            self.call("setAGCmode", 2)
            self.call("disableALSfilter")
            self.call("disableNoiseBlanker")
            self.call("setInputGain", 1.0)
            self.call("setOutputGain", 0.5)
            self.call("setIQgainBalance", 1.02)
            self.call("enableAudioFilter")
            self.call("setAudioFilter", 6)
            self.call("setDemodMode", 0)

    def call(self, name, *args):
        if name in FLOAT_ARG:
            self.r.cpu.call(METHODS[name], [SDR], sargs=[float(args[0])])
        else:
            self.r.call_addr(METHODS[name], SDR, *args)
        if name == "setDemodMode":
            self.tuning_offset = self.r.cpu.fs(0)
        return self

    def poke_u8(self, off, v):
        self.r.m.write(SDR + off, 1, v)

    def obj(self, off, n, dt):
        return self.r.get(SDR + off, n, dt)

    def glob(self, name, n=1, dt=np.float32):
        return self.r.get(GLOBALS[name], n, dt)

    def update(self, i128, q128):
        self.cur[0], self.cur[1] = self.bl.new(i128), self.bl.new(q128)
        self.sent.clear()
        self.r.call_addr(METHODS["update"], SDR)
        return self.sent[0].copy()

    def run(self, i, q):
        return np.concatenate([self.update(i[b * 128:(b + 1) * 128], q[b * 128:(b + 1) * 128]) for b in range(len(i) // 128)])

    def tap(self, name):
        return np.stack(self.taps[name])                      # [blocks][buffers][128]


class PreRef:
    """`AudioSDRpreProcessor preProcessor;` (INO:53): the object as the sketch's static initialiser leaves it
    (ITCM 0x93e8 ... 0x9442: AudioStream's constructor, the vtable, a float 1.0 at +0x420, a flag at +1073, everything else
    zero), then startAutoI2SerrorDetection() (INO:117) unless told otherwise; update() = 0xee88"""
    OBJ = 0x2001dd0c
    UPDATE, START = 0xee88, 0xf084

    def __init__(self, im=None, start=True):
        self.r = r = M.Ref(im or Image())
        self.bl = M.Blocks(r, 16)
        self.cur, self.sent = {}, {}
        h = r.cpu.hooks
        h[M.A["receiveWritable"]] = lambda c: c.r.__setitem__(0, self.cur[c.r[1]])
        h[M.A["release"]] = lambda c: None
        h[M.A["transmit"]] = lambda c: self.sent.__setitem__(c.r[2], r.get(c.r[1] + 4, 128, np.int16))
        r.m.write_bytes(self.OBJ, bytes(1080))
        r.put(self.OBJ + 0x420, np.array([1.0], np.float32))
        r.m.write(self.OBJ + 1073, 1, 1)
        if start:
            r.call_addr(self.START, self.OBJ)

    def swapIQ(self, on):
        self.r.m.write(self.OBJ + 1072, 1, 1 if on else 0)      # INO:118 (commented out in the sketch; the flag update() reads)

    def state(self):
        """slip (-1, 0, 1), bad count, checked blocks, detecting"""
        s = self.r.get(self.OBJ + 1064, 4, np.int16)
        return int(s[0]), int(s[2]), int(s[3]), int(self.r.m.read(self.OBJ + 1074, 1))

    def update(self, i128, q128):
        self.cur[0], self.cur[1] = self.bl.new(i128), self.bl.new(q128)
        self.sent.clear()
        self.r.call_addr(self.UPDATE, self.OBJ)
        return self.sent[0].copy(), self.sent[1].copy()
