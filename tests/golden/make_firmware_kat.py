#!/usr/bin/env python3
"""Known answers computed BY THE REFERENCE'S OWN COMPILED CODE.

The reference ships one artefact that holds its hot path in executable form: pre_compiled/RadioDSP_SDR_RX.ino.hex, a
Teensy 4 (Cortex-M7) build of the sketch with CMSIS-DSP and the Teensy Audio library linked in.  It cannot be built
here and there is no ARM machine, but its routines are plain Thumb-2 code working on memory: tests/golden/thumb_emu.py
interprets that instruction set, tests/golden/firmware_image.py lays the image out as the start-up code does, and this
script calls single routines of it -- by the addresses the image itself calls them at -- on seeded inputs and writes
inputs and outputs to tests/golden/firmware_kat.npz.  Data only: nothing of the image is copied.

What is run (ITCM address; how it was identified):
  arm_cfft_radix4_init_q15 0x12584, arm_cfft_radix4_q15 0x12548   called from both analysers' update()
  arm_lms_norm_init_f32 0x1262c, arm_lms_norm_f32 0x12668         called from Init_LMS_NR / LMS_NoiseReduction (NR:62,73)
  arm_biquad_cascade_df1_init_f32 0x128b0, ..._df1_f32 0x128cc    the engine's audio filters
  arm_q15_to_float 0x116ac, arm_float_to_q15 0x1174c              CONV:241-242, :346-347
  arm_cfft_f32 0x11f68, arm_cmplx_mult_cmplx_f32 0x12a64          CONV:290, :299, :307
  Init_LMS_NR 0x6b74, LMS_NoiseReduction 0x6c0c                   NR:35, NR:66
  calc_cplx_FIR_coeffs 0x6ce0, init_filter_mask 0x6c68, doConvolutionalInitialize 0x704c, reInitializeFilter 0x7094,
  doConvolutionalProcessing 0x70d0                                CONV:127, :87, :187, :209, :228
  AudioAnalyzeFFT256IQ::update 0x9a20, AudioAnalyzeFFT1024::update 0xbb18, AudioFilterBiquad::update 0xc09c and
  ::setCoefficients 0xc17c                                        through the objects' vtables / the sketch's calls
  AudioSDR's constructor 0x6744, ::setDemodMode 0xd798, ::setAudioFilter 0xd97c   INO:138-139, CTL:153-177,330-423 (answers only)
The AudioStream plumbing the update() methods call (receiveReadOnly / receiveWritable / transmit / release) and the
record / play queues of doConvolutionalProcessing are replaced by hooks that hand over our buffers; everything else
-- newlib's powf / sin / cos included -- is the image's code.

Build container only: needs /root/reference.  Run: python tests/golden/make_firmware_kat.py            (writes the fixture)
                                                      python tests/golden/make_firmware_kat.py --check    (recomputes everything
and compares with the committed fixture array by array; exit status 1 on any difference)
"""
import hashlib
import os
import struct
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from firmware_image import SCRATCH, Image  # noqa: E402
from make_firmware_tables import HEX, TABLES  # noqa: E402

OUT = os.path.join(HERE, "firmware_kat.npz")
F32, I16 = np.float32, np.int16

# ITCM addresses of the routines (this image only: its SHA-256 is checked below)
A = dict(cfft_q15_init=0x12584, cfft_q15=0x12548, lms_init=0x1262c, lms=0x12668, df1_init=0x128b0, df1=0x128cc,
         q15_to_float=0x116ac, float_to_q15=0x1174c, cfft_f32=0x11f68, cmplx_mult=0x12a64,
         Init_LMS_NR=0x6b74, LMS_NoiseReduction=0x6c0c, calc_cplx_FIR_coeffs=0x6ce0, init_filter_mask=0x6c68,
         doConvolutionalInitialize=0x704c, reInitializeFilter=0x7094, doConvolutionalProcessing=0x70d0,
         fft256iq_update=0x9a20, fft1024_update=0xbb18, biquad_update=0xc09c, biquad_setCoefficients=0xc17c,
         receiveReadOnly=0x10e04, receiveWritable=0x10e1c, transmit=0x10dd0, release=0x10d7c,
         rq_available=0xca6c, rq_readBuffer=0xcad0, rq_freeBuffer=0xcb0c, rq_begin=0xca84, pq_getBuffer=0xc9d0, pq_playBuffer=0xc9f4)
# globals of the sketch (DTCM), read out of the literal pools of the routines above
G = dict(Q_in_L=0x2001d990, Q_in_R=0x2001a208, Q_out_L=0x2001c430, Q_out_R=0x2001ed48, FIR_Coef_I=0x2001c598, FIR_Coef_Q=0x2001cba8,
         FIR_filter_mask=0x2001a570, lms_instance=0x2001f2fc, lms_coeffs=0x20019e78, cfft_instance_256=0x20003bf0,
         FFT_length=0x200089b8, N_BLOCKS=0x20015a28, m_NumTaps=0x20008e08, first_block=0x20008e04)


def synth_iq(n, seed):
    """two tones + a weak carrier + noise on both rails, int16 [n, 2] (any signal would do: it only has to be the same
    on both sides of the comparison)"""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    ph = 2 * np.pi * (1000.0 / 44100.0) * t
    ph2 = 2 * np.pi * (2600.0 / 44100.0) * t
    ph3 = 2 * np.pi * (-5200.0 / 44100.0) * t
    i = 0.30 * np.cos(ph) + 0.12 * np.cos(ph2) + 0.05 * np.cos(ph3) + 0.02 * rng.standard_normal(n)
    q = 0.30 * np.sin(ph) + 0.12 * np.sin(ph2) + 0.05 * np.sin(ph3) + 0.02 * rng.standard_normal(n)
    return np.stack([np.clip(np.round(i * 32768), -32768, 32767), np.clip(np.round(q * 32768), -32768, 32767)], 1).astype(I16)


class Ref:
    """one emulated machine with the sketch's start-up state"""

    def __init__(self, im):
        self.im = im
        self.cpu = im.machine()
        self.m = self.cpu.mem
        self.m.map(0xE000E000, 0x1000)              # NVIC registers reInitializeFilter writes (interrupt enable bits): plain memory here
        self.next = SCRATCH

    def alloc(self, n, align=8):
        a = (self.next + align - 1) & ~(align - 1)
        self.next = a + n
        assert self.next < SCRATCH + 0x80000
        return a

    def put(self, addr, arr):
        self.m.write_bytes(addr, np.ascontiguousarray(arr).tobytes())

    def get(self, addr, n, dt):
        return np.frombuffer(self.m.read_bytes(addr, n * np.dtype(dt).itemsize), dtype=dt).copy()

    def call(self, name, *args, **kw):
        return self.cpu.call(A[name], list(args), **kw)

    def call_addr(self, addr, *args, **kw):
        return self.cpu.call(addr, list(args), **kw)


# ---- CMSIS leaf routines -------------------------------------------------------------------------------------------------
def kat_cfft_q15(im, out):
    rng = np.random.default_rng(101)
    for n in (256, 1024):
        r = Ref(im)
        S, buf = r.alloc(16), r.alloc(4 * n)
        assert r.call("cfft_q15_init", S, n, 0, 1) == 0              # as the analysers' constructors do
        xs, ys = [], []
        for trial in range(9):
            x = rng.integers(-32768, 32768, 2 * n).astype(I16)
            if trial == 1:
                x = np.where(rng.random(2 * n) < 0.5, 32767, -32768).astype(I16)
            if trial == 2:
                x = rng.integers(-300, 300, 2 * n).astype(I16)
            if trial == 4:                                               # full-scale DC on both rails: every sum saturates or halves
                x = np.full(2 * n, 32767, I16)
            if trial == 5:
                x = np.full(2 * n, -32768, I16)
            if trial == 6:                                               # Nyquist alternation at full scale
                x = np.tile(np.array([32767, -32768, -32768, 32767], I16), n // 2)
            if trial == 7:                                               # one full-scale impulse
                x = np.zeros(2 * n, I16); x[2 * 37] = -32768; x[2 * 37 + 1] = 32767
            if trial == 8:                                               # a full-scale complex tone exactly on a bin
                k = np.arange(n)
                x = np.stack([np.round(32767 * np.cos(2 * np.pi * 19 * k / n)), np.round(32767 * np.sin(2 * np.pi * 19 * k / n))], 1).astype(I16).reshape(-1)
            r.put(buf, x)
            r.call("cfft_q15", S, buf)
            xs.append(x)
            ys.append(r.get(buf, 2 * n, I16))
        out[f"cfft_q15_{n}_in"], out[f"cfft_q15_{n}_out"] = np.stack(xs), np.stack(ys)


def kat_lms_norm(im, out):
    """arm_lms_norm_f32 as NR:62,73 use it: 96 taps, blocks of 128, three consecutive blocks on one instance"""
    rng = np.random.default_rng(102)
    r = Ref(im)
    taps, n, nblk = 96, 128, 3
    S, co, st = r.alloc(24), r.alloc(4 * taps), r.alloc(4 * (taps + n))
    src, ref, o, e = (r.alloc(4 * n) for _ in range(4))
    c0 = (rng.standard_normal(taps) * 0.03).astype(F32)
    mu = F32(0.1122018)
    r.put(co, c0)
    r.call("lms_init", S, taps, co, st, sargs=[float(mu)], stack_args=[n])
    xs = (rng.standard_normal((nblk, n)) * 0.2).astype(F32)
    ds = (0.7 * xs + rng.standard_normal((nblk, n)) * 0.05).astype(F32)
    outs, errs, cos_, inst = [], [], [], []
    for b in range(nblk):
        r.put(src, xs[b]); r.put(ref, ds[b])
        r.call("lms", S, src, ref, o, stack_args=[e, n])
        outs.append(r.get(o, n, F32)); errs.append(r.get(e, n, F32)); cos_.append(r.get(co, taps, F32))
        inst.append(r.get(S + 16, 2, F32))                            # energy, x0
    out.update(lms_taps=np.int64(taps), lms_mu=np.array([mu], F32), lms_coeffs0=c0, lms_src=xs, lms_ref=ds, lms_out=np.stack(outs),
               lms_err=np.stack(errs), lms_coeffs=np.stack(cos_), lms_energy_x0=np.stack(inst))


def kat_df1(im, out, biquad_sets):
    """arm_biquad_cascade_df1_f32 with the engine's own first set (four sections), 2 x 160 samples on one instance"""
    rng = np.random.default_rng(103)
    r = Ref(im)
    n = 160
    S, co, st, src, dst = r.alloc(12), r.alloc(80), r.alloc(64), r.alloc(4 * n), r.alloc(4 * n)
    coef = np.ascontiguousarray(biquad_sets[0].reshape(20), F32)
    r.put(co, coef)
    r.call("df1_init", S, 4, co, st)
    xs = (rng.standard_normal((2, n)) * 0.25).astype(F32)
    ys = []
    for b in range(2):
        r.put(src, xs[b])
        r.call("df1", S, src, dst, n)
        ys.append(r.get(dst, n, F32))
    out.update(df1_coef=coef, df1_in=xs, df1_out=np.stack(ys))


def kat_converters(im, out):
    rng = np.random.default_rng(104)
    r = Ref(im)
    n = 203                                                            # not a multiple of four: the tail loop runs too
    a, b = r.alloc(4 * n), r.alloc(4 * n)
    x = (rng.standard_normal(n) * 0.4).astype(F32)
    k = np.array([0.4, 0.5, 0.6, 1.5, 2.5, -0.4, -0.5, -0.6, -1.5, -2.5, 32766.5, 32767.5, -32768.5, 40000.0, -40000.0, 0.0], F32) / F32(32768.0)
    x[:len(k)] = k
    r.put(a, x)
    r.call("float_to_q15", a, b, n)
    out.update(float_to_q15_in=x, float_to_q15_out=r.get(b, n, I16))
    q = rng.integers(-32768, 32768, n).astype(I16)
    q[:3] = (-32768, 32767, 0)
    r.put(a, q)
    r.call("q15_to_float", a, b, n)
    out.update(q15_to_float_in=q, q15_to_float_out=r.get(b, n, F32))


def kat_cfft_f32(im, out):
    """arm_cfft_f32 with the instance the sketch uses (arm_cfft_sR_f32_len256), forward and inverse, bit reversal on;
    arm_cmplx_mult_cmplx_f32"""
    rng = np.random.default_rng(105)
    r = Ref(im)
    n = 256
    buf, b2, b3 = r.alloc(8 * n), r.alloc(8 * n), r.alloc(8 * n)
    x = (rng.standard_normal(2 * n) * 0.3).astype(F32)
    r.put(buf, x)
    r.call("cfft_f32", G["cfft_instance_256"], buf, 0, 1)
    fwd = r.get(buf, 2 * n, F32)
    r.call("cfft_f32", G["cfft_instance_256"], buf, 1, 1)
    out.update(cfft_f32_in=x, cfft_f32_fwd=fwd, cfft_f32_inv=r.get(buf, 2 * n, F32))
    y = (rng.standard_normal(2 * n) * 0.3).astype(F32)
    r.put(buf, x); r.put(b2, y)
    r.call("cmplx_mult", buf, b2, b3, n)
    out.update(cmplx_mult_a=x, cmplx_mult_b=y, cmplx_mult_out=r.get(b3, 2 * n, F32))


# ---- the sketch's own routines -----------------------------------------------------------------------------------------------
class Sketch(Ref):
    """the CONV stage as .ino:172-198 runs it: Init_LMS_NR(15); doConvolutionalInitialize(); reInitializeFilter(300, 4000);
    then doConvolutionalProcessing(nr_level, true, 300.0, 4000.0) once per audio block"""

    def __init__(self, im):
        super().__init__(im)
        self.buf = {G["Q_in_L"]: self.alloc(256), G["Q_in_R"]: self.alloc(256), G["Q_out_L"]: self.alloc(256), G["Q_out_R"]: self.alloc(256)}
        h = self.cpu.hooks

        def ret(v):
            def f(c): c.r[0] = v
            return f

        def buf_of(c): c.r[0] = self.buf[c.r[0]]
        h[A["rq_begin"]] = ret(0)
        h[A["rq_available"]] = ret(2)                                  # "> N_BLOCKS" with N_BLOCKS = 1
        h[A["rq_readBuffer"]] = buf_of
        h[A["rq_freeBuffer"]] = ret(0)
        h[A["pq_getBuffer"]] = buf_of
        h[A["pq_playBuffer"]] = ret(0)
        self.prepack = []
        self.cpu.watch[A["float_to_q15"]] = lambda c: self.prepack.append(self.get(c.r[0], c.r[2], F32))
        assert self.m.read(G["FFT_length"], 4) == 256 and self.m.read(G["N_BLOCKS"], 4) == 1 and self.m.read(G["m_NumTaps"], 4) == 129

    def setup(self, nr_init=15, lo=300.0, hi=4000.0):
        self.call("Init_LMS_NR", nr_init)                               # INO:172
        self.call("doConvolutionalInitialize")                          # INO:180
        self.call("reInitializeFilter", dargs={0: lo, 1: hi})            # INO:183

    def taps(self):
        return self.get(G["FIR_Coef_I"], 129, np.float64), self.get(G["FIR_Coef_Q"], 129, np.float64)

    def mask(self):
        return self.get(G["FIR_filter_mask"], 512, F32)

    def process(self, iq, nr, filt=1):
        """iq int16 [n, 2] -> (int16 [n, 2], float32 [n, 2] as handed to arm_float_to_q15)"""
        o16, o32 = [], []
        for b in range(len(iq) // 128):
            blk = iq[b * 128:(b + 1) * 128]
            self.put(self.buf[G["Q_in_L"]], blk[:, 0]); self.put(self.buf[G["Q_in_R"]], blk[:, 1])
            self.prepack.clear()
            self.call("doConvolutionalProcessing", int(filt), sargs=[float(nr)], dargs={1: 300.0, 2: 4000.0})
            assert len(self.prepack) == 2
            o32.append(np.stack(self.prepack, 1))
            o16.append(np.stack([self.get(self.buf[G["Q_out_L"]], 128, I16), self.get(self.buf[G["Q_out_R"]], 128, I16)], 1))
        return np.concatenate(o16), np.concatenate(o32)


def kat_design(im, out):
    """calc_cplx_FIR_coeffs + init_filter_mask through reInitializeFilter for the sketch's pass band and three others
    (the PBT range of CTL:569-612); Init_LMS_NR's mu for every strength"""
    bands = [(300.0, 4000.0), (150.0, 2400.0), (700.0, 800.0), (0.0, 4000.0)]
    ti, tq, mk = [], [], []
    for lo, hi in bands:
        s = Sketch(im)
        s.setup(15, lo, hi)
        a, b = s.taps()
        ti.append(a); tq.append(b); mk.append(s.mask())
    out.update(design_bands=np.array(bands), design_taps_i=np.stack(ti), design_taps_q=np.stack(tq), design_mask=np.stack(mk))
    s = Sketch(im)
    mus = []
    for strength in range(1, 41):
        s.call("Init_LMS_NR", strength)
        mus.append(s.get(G["lms_instance"] + 12, 1, F32)[0])
    out["init_lms_nr_mu"] = np.array(mus, F32)


def kat_lms_noise_reduction(im, out):
    """Init_LMS_NR(20) + LMS_NoiseReduction(128, buffer) (NR:35-80: the 256-float delay ring around arm_lms_norm_f32) on 24
    float blocks -- row A7 by itself, no transform in front of it"""
    rng = np.random.default_rng(106)
    s = Sketch(im)
    s.call("Init_LMS_NR", 20)
    n = 24 * 128
    t = np.arange(n)
    x = (0.2 * np.sin(2 * np.pi * 0.031 * t) + 0.1 * np.sin(2 * np.pi * 0.113 * t + 1.0) + 0.05 * rng.standard_normal(n)).astype(F32)
    buf = s.alloc(512)
    ys = []
    for b in range(24):
        s.put(buf, x[b * 128:(b + 1) * 128])
        s.call("LMS_NoiseReduction", 128, buf)
        ys.append(s.get(buf, 128, F32))
    out.update(lmsnr_strength=np.int64(20), lmsnr_in=x, lmsnr_out=np.concatenate(ys), lmsnr_coeffs=s.get(G["lms_coeffs"], 96, F32))


def kat_conv(im, out):
    """doConvolutionalProcessing block by block on one stream each:
       conv_plain   nr 0                              32 blocks  (A1 unpack, A5 overlap-save filter, A10 pack)
       conv_nr15    nr 15 (the sketch's start-up)     48 blocks  (+ A7 NLMS noise reduction, x 1.1, L copied to R)
       conv_nrstep  nr 15 -> 30 after 16 blocks       32 blocks  (Init_LMS_NR in mid-stream: state cleared, taps kept)
       conv_pbt     nr 0, reInitializeFilter(450, 2700) after 12 blocks, 24 blocks
       conv_nofilt  bFilterEnabled false              12 blocks  (CONV:303 copies FFT_length floats = half the spectrum)
       conv_loud    nr 15, input at 3.2 x the level    16 blocks  (rails at the input, arm_float_to_q15 saturating at the output)
       conv_fade    nr 15, 12 blocks at 2.5 x, then 28 blocks 48 dB down (arm_lms_norm_f32's running energy after a loud passage)
    and the number of instructions the image executes per block (plain / with the NLMS) -- what the reference's own
    processor has to get through in the 2.9 ms a block lasts"""
    t0 = time.time()
    iq = synth_iq(48 * 128, 7)
    s = Sketch(im); s.setup()
    c0 = s.cpu.count
    o16, o32 = s.process(iq[:32 * 128], 0.0)
    out.update(conv_iq=iq, conv_plain_o16=o16, conv_plain_o32=o32, conv_plain_instructions_per_block=np.int64((s.cpu.count - c0) // 32))
    s = Sketch(im); s.setup()
    c0 = s.cpu.count
    o16, o32 = s.process(iq, 15.0)
    out.update(conv_nr15_o16=o16, conv_nr15_o32=o32, conv_nr15_coeffs=s.get(G["lms_coeffs"], 96, F32),
               conv_nr15_instructions_per_block=np.int64((s.cpu.count - c0) // 48))
    loud = np.clip(np.round(iq[:16 * 128].astype(np.float64) * 3.2), -32768, 32767).astype(I16)
    s = Sketch(im); s.setup()
    o16, o32 = s.process(loud, 15.0)
    out.update(conv_loud_iq=loud, conv_loud_o16=o16, conv_loud_o32=o32)
    g = np.where(np.arange(40 * 128) < 12 * 128, 2.5, 0.01)[:, None]
    fade = np.clip(np.round(iq[:40 * 128].astype(np.float64) * g), -32768, 32767).astype(I16)
    s = Sketch(im); s.setup()
    o16, o32 = s.process(fade, 15.0)
    out.update(conv_fade_iq=fade, conv_fade_o16=o16, conv_fade_o32=o32, conv_fade_energy_x0=s.get(G["lms_instance"] + 16, 2, F32))
    s = Sketch(im); s.setup()
    a16, a32 = s.process(iq[:16 * 128], 15.0)
    b16, b32 = s.process(iq[16 * 128:32 * 128], 30.0)
    out.update(conv_nrstep_o16=np.concatenate([a16, b16]), conv_nrstep_o32=np.concatenate([a32, b32]))
    s = Sketch(im); s.setup()
    a16, a32 = s.process(iq[:12 * 128], 0.0)
    s.call("reInitializeFilter", dargs={0: 450.0, 1: 2700.0})
    b16, b32 = s.process(iq[12 * 128:24 * 128], 0.0)
    out.update(conv_pbt_o16=np.concatenate([a16, b16]), conv_pbt_o32=np.concatenate([a32, b32]))
    s = Sketch(im); s.setup()
    o16, o32 = s.process(iq[:12 * 128], 0.0, filt=0)
    out.update(conv_nofilt_o16=o16, conv_nofilt_o32=o32)
    print("  conv runs: %.0f s" % (time.time() - t0))


# ---- the graph's objects ---------------------------------------------------------------------------------------------------
class Blocks:
    """audio_block_t buffers (ref count, pool index, int16 data[128] at +4) handed to update() by the hooks"""

    def __init__(self, ref, n=64):
        self.ref = ref
        self.slots = [ref.alloc(260, 4) for _ in range(n)]
        self.k = 0

    def new(self, data=None):
        a = self.slots[self.k % len(self.slots)]
        self.k += 1
        self.ref.m.write_bytes(a, bytes(4))
        self.ref.put(a + 4, np.zeros(128, I16) if data is None else np.ascontiguousarray(data, I16))
        return a


def kat_fft256iq(im, out, tables):
    """AudioAnalyzeFFT256IQ::update on 2 x 40 blocks: object memory zeroed, then the constructor's settings (FFTIQ.h:55-60:
    window, naverage, arm_cfft_radix4_init_q15 run from the image) and the sketch's (INO:144-145: Hanning, averageTogether(30));
    every time outputflag comes up the 256 output words are recorded"""
    iq = synth_iq(40 * 128, 11)
    iq[5 * 128:6 * 128] = np.where(np.random.default_rng(1).random((128, 2)) < 0.5, 32767, -32768)   # one block on the rails
    out["fft256iq_iq"] = iq
    k = np.arange(6 * 128)
    adv = np.concatenate([
        np.tile(np.array([[32767, -32768]], I16), (6 * 128, 1)),                                           # full-scale DC
        np.stack([np.where(k % 2 == 0, 32767, -32768), np.where(k % 2 == 0, -32768, 32767)], 1).astype(I16),   # Nyquist alternation
        np.stack([np.round(32767 * np.cos(2 * np.pi * 19 * k / 256)), np.round(32767 * np.sin(2 * np.pi * 19 * k / 256))], 1).astype(I16),  # a tone on a bin
        np.random.default_rng(4).integers(-32768, 32768, (6 * 128, 2)).astype(I16)])                        # full-scale noise
    out["fft256iq_adv_iq"] = adv
    for tag, window, navg in (("sketch", "hann256", 30), ("default", "blackman_nuttall256", 8), ("avg1", "hann256", 1), ("adv", "hann256", 2)):
        if tag == "adv":
            iq = adv
        r = Ref(im)
        obj = r.alloc(0xa40 + 16)
        r.m.write(obj + 536, 4, im.dtcm_of_offset(TABLES[window][0]))
        r.m.write(obj + 2597, 1, navg)
        assert r.call("cfft_q15_init", obj + 0xa30, 256, 0, 1) == 0
        bl = Blocks(r)
        cur = {}
        r.cpu.hooks[A["receiveReadOnly"]] = lambda c: c.r.__setitem__(0, cur[c.r[1]])
        r.cpu.hooks[A["release"]] = lambda c: None
        spectra, ticks = [], []
        for b in range(len(iq) // 128):
            blk = iq[b * 128:(b + 1) * 128]
            cur[0], cur[1] = bl.new(blk[:, 0]), bl.new(blk[:, 1])
            r.call("fft256iq_update", obj)
            if r.m.read(obj + 2598, 1):
                r.m.write(obj + 2598, 1, 0)                              # available() clears it
                spectra.append(r.get(obj + 24, 256, np.uint16)); ticks.append(b)
        out[f"fft256iq_{tag}_out"] = np.stack(spectra)
        out[f"fft256iq_{tag}_ticks"] = np.array(ticks)


def kat_fft1024(im, out, tables):
    """AudioAnalyzeFFT1024::update (Teensy Audio library, `AudioFFT` on Q_out_L, INO:57,147) on 40 blocks with
    AudioWindowHanning1024 and with no window"""
    x = synth_iq(40 * 128, 12)[:, 0].copy()
    x[9 * 128:10 * 128] = np.where(np.random.default_rng(2).random(128) < 0.5, 32767, -32768)
    out["fft1024_in"] = x
    for tag, window in (("hann", "hann1024"), ("nowindow", None)):
        r = Ref(im)
        obj = r.alloc(0x1444 + 32)
        r.m.write(obj + 1048, 4, im.dtcm_of_offset(TABLES[window][0]) if window else 0)
        assert r.call("cfft_q15_init", obj + 0x1444, 1024, 0, 1) == 0
        bl = Blocks(r)
        cur = {}
        r.cpu.hooks[A["receiveReadOnly"]] = lambda c: c.r.__setitem__(0, cur[0])
        r.cpu.hooks[A["release"]] = lambda c: None
        spectra, ticks = [], []
        for b in range(len(x) // 128):
            cur[0] = bl.new(x[b * 128:(b + 1) * 128])
            r.call("fft1024_update", obj)
            if r.m.read(obj + 0x143d, 1):
                r.m.write(obj + 0x143d, 1, 0)
                spectra.append(r.get(obj + 24, 512, np.uint16)); ticks.append(b)
        out[f"fft1024_{tag}_out"] = np.stack(spectra)
        out[f"fft1024_{tag}_ticks"] = np.array(ticks)


def kat_panadapter(im, out, tables):
    """The panadapter branch of the sketch's graph, end to end from the image (INO:57-60,75-78,144-145,155-156):
    IQinput -> biquad1 / biquad2 (setHighpass(0, 500, 0.5): the integers setup() holds) -> FFT (AudioWindowHanning256,
    averageTogether(30)); 96 blocks, three spectra.  Each update() is the image's; AudioStream's hand-over between the
    nodes is done by the hooks (biquad output blocks become the analyser's input blocks)."""
    iq = synth_iq(96 * 128, 14)
    r = Ref(im)
    coef = r.alloc(20)
    r.put(coef, teensy_biquad_ints("highpass", 500.0, 0.5))
    bq = [r.alloc(24 + 32 * 4 + 16) for _ in range(2)]
    for o in bq:
        r.call("biquad_setCoefficients", o, 0, coef)
    fft = r.alloc(0xa40 + 16)
    r.m.write(fft + 536, 4, im.dtcm_of_offset(TABLES["hann256"][0]))
    r.m.write(fft + 2597, 1, 30)
    assert r.call("cfft_q15_init", fft + 0xa30, 256, 0, 1) == 0
    bl = Blocks(r, 16)
    cur, sent = {}, {}
    r.cpu.hooks[A["receiveWritable"]] = lambda c: c.r.__setitem__(0, cur["w"])
    r.cpu.hooks[A["transmit"]] = lambda c: sent.__setitem__("b", c.r[1])
    r.cpu.hooks[A["receiveReadOnly"]] = lambda c: c.r.__setitem__(0, cur[c.r[1]])
    r.cpu.hooks[A["release"]] = lambda c: None
    spectra = []
    for b in range(len(iq) // 128):
        blk = iq[b * 128:(b + 1) * 128]
        for side in (0, 1):
            cur["w"] = bl.new(blk[:, side])
            r.call("biquad_update", bq[side])
            cur[side] = sent["b"]
        r.call("fft256iq_update", fft)
        if r.m.read(fft + 2598, 1):
            r.m.write(fft + 2598, 1, 0)
            spectra.append(r.get(fft + 24, 256, np.uint16))
    out.update(panadapter_iq=iq, panadapter_out=np.stack(spectra))


def teensy_biquad_ints(kind, frequency, q, fs=44100.0):
    """filter_biquad.h's setters as published (all in double, x 2^30, int conversion): gives the five integers the image's
    setup() holds for setHighpass(0, 500, 0.5)"""
    w0 = float(F32(frequency)) * (2.0 * 3.141592654 / fs)
    sw, cw = np.sin(w0), np.cos(w0)
    alpha = sw / (float(F32(q)) * 2.0)
    scale = 1073741824.0 / (1.0 + alpha)
    if kind == "lowpass": b = [((1.0 - cw) / 2.0) * scale, (1.0 - cw) * scale]; b.append(b[0])
    elif kind == "highpass": b = [((1.0 + cw) / 2.0) * scale, -(1.0 + cw) * scale]; b.append(b[0])
    elif kind == "bandpass": b = [alpha * scale, 0.0, (-alpha) * scale]
    else: b = [scale, (-2.0 * cw) * scale]; b.append(b[0])
    c = b + [(-2.0 * cw) * scale, (1.0 - alpha) * scale]
    return np.array([int(v) for v in c], np.int32)


def kat_teensy_biquad(im, out):
    """AudioFilterBiquad: setCoefficients + update from the image on one object per case:
       hp      stage 0 = setHighpass(0, 500, 0.5) (INO:155)
       chain   stages 0, 1, 2 = high-pass, notch, low-pass
       gap     stages 0 and 2 set, 1 never: update() must stop after stage 0
       fresh   nothing set: passes nothing
    32 blocks each; blocks 4-5 are full-scale noise (the saturating path)"""
    x = synth_iq(32 * 128, 13)[:, 1].copy()
    x[4 * 128:6 * 128] = np.random.default_rng(3).integers(-32768, 32768, 256).astype(I16)
    out["tbq_in"] = x
    cases = dict(hp=[(0, "highpass", 500.0, 0.5)], chain=[(0, "highpass", 500.0, 0.5), (1, "notch", 1000.0, 4.0), (2, "lowpass", 3000.0, 0.707)],
                 gap=[(0, "lowpass", 2000.0, 0.8), (2, "highpass", 300.0, 0.7)], fresh=[])
    for tag, stages in cases.items():
        r = Ref(im)
        obj = r.alloc(24 + 32 * 4 + 16)
        cbuf = r.alloc(20)
        coefs = np.zeros((4, 5), np.int32)
        for st, kind, f, q in stages:
            coefs[st] = teensy_biquad_ints(kind, f, q)
            r.put(cbuf, coefs[st])
            r.call("biquad_setCoefficients", obj, st, cbuf)
        bl = Blocks(r)
        cur, sent = {}, []
        r.cpu.hooks[A["receiveWritable"]] = lambda c: c.r.__setitem__(0, cur[0])
        r.cpu.hooks[A["transmit"]] = lambda c: sent.append(r.get(c.r[1] + 4, 128, I16))
        r.cpu.hooks[A["release"]] = lambda c: None
        for b in range(len(x) // 128):
            cur[0] = bl.new(x[b * 128:(b + 1) * 128])
            r.call("biquad_update", obj)
        out[f"tbq_{tag}_coefs"] = coefs
        out[f"tbq_{tag}_stages"] = np.array([s[0] for s in stages], np.int64)
        out[f"tbq_{tag}_out"] = np.concatenate(sent)


def kat_engine(im, out, tables):
    """What the un-vendored AudioSDR engine answers to the two calls of the sketch's mode menu (CTL:330-423), asked of the
    engine itself: its constructor (0x6744: AudioStream set-up, the defaults, fifteen coefficient sets copied into the
    object) is run on zeroed memory, then
      * setDemodMode(mode) (0xd798) for LSBmode ... SAMmode = 0 ... 5 (the numbers the compiled tuningMode() passes):
        the TuningOffset it returns (INO:139, CTL:337-407) -- the engine is a low-IF receiver: centre 6890 Hz, SSB band
        3000 Hz, CW band 1000 Hz (the floats at +32 / +36 / +40 of the object), carrier at centre +- half the band;
      * setAudioFilter(k) (0xd97c) for k = 0 ... 9: which of the image's fifteen sets of four biquad sections
        (firmware_tables.npz `biquad_sets`) it installs; the compiled filterMode() / tuningMode() pass audioAM = 0,
        audioCW = 1, audio2100 = 3, audio2700 = 6, audio3100 = 8; and which sets setDemodMode installs in front of the
        demodulator (the band-passes around the IF)."""
    r = Ref(im)
    sdr = 0x20017208                                                      # where the sketch's `SDR` object lives (.bss)
    r.call_addr(0x6744, sdr)
    sets = tables["biquad_sets"].reshape(15, 20)
    out["engine_if_centre_ssb_cw"] = r.get(sdr + 32, 3, F32)
    offs = []
    for mode in range(6):
        r.call_addr(0xd798, sdr, mode)
        offs.append(r.cpu.fs(0))
    out["engine_tuning_offset"] = np.array(offs, F32)
    out["engine_demod_names"] = np.array(["LSBmode", "USBmode", "CW_LSBmode", "CW_USBmode", "AMmode", "SAMmode"])

    def which(off):
        a = r.get(sdr + off, 20, F32)
        hit = [i for i in range(15) if np.array_equal(a, sets[i])]
        assert len(hit) == 1
        return hit[0]
    aud = {0: 0x24d8, 1: 0x2528, 2: 0x2578, 3: 0x22a8, 4: 0x22f8, 5: 0x2348, 6: 0x2398, 7: 0x23e8, 8: 0x2438, 9: 0x2488}   # setAudioFilter's own switch
    out["engine_audio_filter_set"] = np.array([which(aud[k]) for k in range(10)], np.int64)
    out["engine_audio_filter_names"] = np.array(["audioAM", "audioCW", "", "audio2100", "", "", "audio2700", "", "audio3100", ""])
    out["engine_if_filter_set"] = np.array([which(0x2668), which(0x2668), which(0x25c8), which(0x25c8), which(0x2708), which(0x2708)], np.int64)
    # the instance setAudioFilter(6) leaves behind points at the set it names
    r.call_addr(0xd97c, sdr, 6)
    assert r.m.read(sdr + 0x688 + 8, 4) == sdr + aud[6]


def _stub_calls(r, code, start, stop, keep=()):
    """every BL inside [start, stop) goes to a no-op, except the targets in `keep`"""
    for o in range(start, stop, 2):
        hw1, hw2 = struct.unpack_from("<HH", code, o)
        if (hw1 & 0xF800) == 0xF000 and (hw2 & 0xD000) == 0xD000:
            s, j1, j2 = (hw1 >> 10) & 1, (hw2 >> 13) & 1, (hw2 >> 11) & 1
            imm = (s << 24) | ((1 - (j1 ^ s)) << 23) | ((1 - (j2 ^ s)) << 22) | ((hw1 & 0x3FF) << 12) | ((hw2 & 0x7FF) << 1)
            if imm & (1 << 24):
                imm -= 1 << 25
            if o + 4 + imm not in keep:
                r.cpu.hooks[o + 4 + imm] = lambda c: c.r.__setitem__(0, 0)


def kat_mode_menu(im, out):
    """tuningMode() (CTL:330-423, ITCM 0x8378) as compiled: for every menu entry mndx = 0 ... 6 and a VFO below and above
    10 MHz the function is run with its display calls stubbed, and what it hands to SDR.setAudioFilter and
    SDR.setDemodMode is recorded (the engine's own numbers: see kat_engine for the names)"""
    code = im.img[im.itcm_off:im.itcm_off + im.etext]
    mndx_addr, vfo_addr = 0x20008dd0, 0x20015a2c                       # literal pool of tuningMode()
    table = np.zeros((7, 2, 2), np.int64)
    for mndx in range(7):
        for k, vfo in enumerate((7030000, 14060000)):
            r = Ref(im)
            got = {}
            _stub_calls(r, code, 0x8378, 0x854c)
            r.cpu.hooks[0xf494] = lambda c: None                       # delay(200), reached by a tail branch
            r.cpu.hooks[0xd97c] = lambda c: got.__setitem__("filter", c.r[1])
            r.cpu.hooks[0xd798] = lambda c: got.__setitem__("mode", c.r[1])
            r.m.write(mndx_addr, 4, mndx)
            r.m.write(vfo_addr, 4, vfo)
            r.call_addr(0x8378)
            table[mndx, k] = (got["filter"], got["mode"])
    out["mode_menu_filter_and_mode"] = table
    out["mode_menu_vfo"] = np.array([7030000, 14060000], np.int64)
    # filterMode() (CTL:149-191, ITCM 0x8084): the audio filter of menu entry fndx = 0 ... 4
    filt = []
    for fndx in range(5):
        r = Ref(im)
        got = {}
        _stub_calls(r, code, 0x8084, 0x8130)
        r.cpu.hooks[0xf494] = lambda c: None
        r.cpu.hooks[0xd97c] = lambda c: got.__setitem__("filter", c.r[1])
        r.m.write(0x20008de4, 4, fndx)
        r.call_addr(0x8084)
        filt.append(got["filter"])
    out["filter_menu_filter"] = np.array(filt, np.int64)


def kat_pbt(im, out):
    """checkPBT_Increase / checkPBT_Decrease (CTL:569-612; ITCM 0x8a48, 0x8b08) as compiled: menu level L4_PBT_LH, one
    of the two buttons held (digitalRead answered by a hook), reInitializeFilter and showPBT stubbed; a walk of 114 presses
    from the sketch's start-up cut-offs up and down into all four limits; the cut-offs after every press"""
    lo_addr, hi_addr, level_addr = 0x20008df8, 0x20008df0, 0x20008e00   # dFLoCut, dFHiCut, iMenuLevel (literal pools)
    r = Ref(im)
    pressed = {}
    r.cpu.hooks[0xf5d4] = lambda c: c.r.__setitem__(0, 0 if c.r[0] == pressed["pin"] else 1)   # digitalRead: LOW = pressed
    calls = []
    r.cpu.hooks[A["reInitializeFilter"]] = lambda c: calls.append((c.fd(0), c.fd(1)))
    r.cpu.hooks[0x7594] = lambda c: None                                  # showPBT
    r.m.write(level_addr, 4, 4)
    start = (r.get(lo_addr, 1, np.float64)[0], r.get(hi_addr, 1, np.float64)[0])
    walk = [(0, +1)] * 12 + [(0, -1)] * 18 + [(1, +1)] * 4 + [(1, -1)] * 70 + [(1, +1)] * 6 + [(0, +1)] * 4      # (edge: 0 LOCUT / D3, 1 HICUT / D6; direction)
    after = []
    for edge, direction in walk:
        pressed["pin"] = 6 if edge == 0 else 3                           # the pin numbers the compiled code asks for, in its order
        assert r.call_addr(0x8a48 if direction > 0 else 0x8b08) == 1
        after.append((r.get(lo_addr, 1, np.float64)[0], r.get(hi_addr, 1, np.float64)[0]))
        assert calls[-1] == after[-1]                                     # reInitializeFilter(dFLoCut, dFHiCut) with the new values
    pressed["pin"] = 99
    assert r.call_addr(0x8a48) == 0 and r.call_addr(0x8b08) == 0           # no button: nothing happens
    out.update(pbt_start=np.array(start), pbt_walk=np.array(walk, np.int64), pbt_after=np.array(after))


class _Stop(Exception):
    pass


def kat_setup(im, out, tables):
    """What the sketch's setup() (INO:140-183, ITCM 0x8f28) hands to the objects of the panadapter branch: the function is
    run with every routine it calls replaced by a no-op, except that AudioFilterBiquad::setCoefficients records its
    arguments -- `biquad1.setHighpass(0, 500, 0.5)` / `biquad2...` (INO:155-156) are inlined and folded to five integer
    literals at compile time -- and that the run ends at the second of them.  By then `FFT.windowFunction(
    AudioWindowHanning256); FFT.averageTogether(30)` (INO:144-145) and `AudioFFT.windowFunction(AudioWindowHanning1024)`
    (INO:147) have written into their objects."""
    r = Ref(im)
    start, stop = 0x8f28, 0x904c
    code = im.img[im.itcm_off:im.itcm_off + im.etext]
    for o in range(start, stop, 2):
        hw1, hw2 = struct.unpack_from("<HH", code, o)
        if (hw1 & 0xF800) == 0xF000 and (hw2 & 0xD000) == 0xD000:
            s, j1, j2 = (hw1 >> 10) & 1, (hw2 >> 13) & 1, (hw2 >> 11) & 1
            imm = (s << 24) | ((1 - (j1 ^ s)) << 23) | ((1 - (j2 ^ s)) << 22) | ((hw1 & 0x3FF) << 12) | ((hw2 & 0x7FF) << 1)
            if imm & (1 << 24):
                imm -= 1 << 25
            r.cpu.hooks[o + 4 + imm] = lambda c: c.r.__setitem__(0, 0)
    got = []

    def set_coefficients(c):
        got.append((c.r[0], c.r[1], r.get(c.r[2], 5, np.int32)))
        if len(got) == 2:
            raise _Stop
    r.cpu.hooks[A["biquad_setCoefficients"]] = set_coefficients
    try:
        r.call_addr(start)
    except _Stop:
        pass
    assert len(got) == 2 and got[0][1] == 0 and got[1][1] == 0 and got[0][0] != got[1][0]
    out["setup_sethighpass_500_05"] = np.stack([g[2] for g in got])
    fft, afft = 0x200167c8, 0x2001ad70                                 # the objects FFT and AudioFFT (literal pool of setup())
    out["setup_fft_naverage"] = np.int64(r.m.read(fft + 2597, 1))
    assert r.m.read(fft + 536, 4) == im.dtcm_of_offset(TABLES["hann256"][0])
    assert r.m.read(afft + 1048, 4) == im.dtcm_of_offset(TABLES["hann1024"][0])
    out["setup_windows"] = np.array(["hann256", "hann1024"])


def main():
    if not os.path.exists(HEX):
        sys.exit("the reference tree is not here: this script runs in the build container only")
    im = Image()
    sha = hashlib.sha256(im.img).hexdigest()
    tables = np.load(os.path.join(HERE, "firmware_tables.npz"))
    out = {"image_sha256": np.array(sha), "image_bytes": np.int64(len(im.img))}
    t0 = time.time()
    for name, f in (("cfft_q15", lambda: kat_cfft_q15(im, out)), ("lms_norm", lambda: kat_lms_norm(im, out)),
                    ("df1", lambda: kat_df1(im, out, tables["biquad_sets"])), ("converters", lambda: kat_converters(im, out)),
                    ("cfft_f32", lambda: kat_cfft_f32(im, out)), ("design", lambda: kat_design(im, out)), ("lms_noise_reduction", lambda: kat_lms_noise_reduction(im, out)),
                    ("conv", lambda: kat_conv(im, out)),
                    ("fft256iq", lambda: kat_fft256iq(im, out, tables)), ("fft1024", lambda: kat_fft1024(im, out, tables)),
                    ("teensy_biquad", lambda: kat_teensy_biquad(im, out)), ("setup", lambda: kat_setup(im, out, tables)),
                    ("engine", lambda: kat_engine(im, out, tables)), ("panadapter", lambda: kat_panadapter(im, out, tables)), ("mode_menu", lambda: kat_mode_menu(im, out)), ("pbt", lambda: kat_pbt(im, out))):
        t = time.time()
        f()
        print("%-14s %.1f s" % (name, time.time() - t), flush=True)
    if "--check" in sys.argv[1:]:
        have = np.load(OUT)
        bad = [k for k in out if k not in have.files or not np.array_equal(np.asarray(out[k]), have[k])]
        bad += [k for k in have.files if k not in out]
        print("check against", OUT, ":", "identical (%d arrays)" % len(out) if not bad else "DIFFERENT: %s" % bad, "-- %.0f s" % (time.time() - t0))
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes, %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
