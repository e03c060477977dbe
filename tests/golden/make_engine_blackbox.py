#!/usr/bin/env python3
"""The un-vendored AudioSDR engine, measured as a black box.

The engine (mixer, side-band selection, audio filters, AGC: SURVEY rows A2, A8, A9 and the engine features of F3) is not
in the reference tree, so this build's versions of those stages are designs of its own.  The engine's compiled code is
in the firmware image, though, and AudioSDR::update() runs under tests/golden/thumb_emu.py like everything else: this
script sets the object up the way the sketch does (INO:117-139), feeds it IQ blocks and records what comes out -- not to
pin anything bit-wise (the algorithms differ), but so that the differences between the engine and this build's
stand-ins are known numbers instead of unknowns.  Results: tests/golden/engine_blackbox.npz, read by
tests/test_firmware_kat.py::test_engine_black_box_facts and quoted in docs/widened_rows.md.

Build container only (needs /root/reference); about six minutes on six cores.
    python tests/golden/make_engine_blackbox.py
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "engine_blackbox.npz")
FS = 44100.0
SDR = 0x20017208                      # the sketch's `SDR` object
E = dict(ctor=0x6744, update=0xe730, enableAGC=0xdfd4, setAGCmode=0xdfe0, disableALSfilter=0xdb14, disableNoiseBlanker=0xe380,
         setInputGain=0xd8a0, setOutputGain=0xd918, setIQgainBalance=0xd8f0, enableAudioFilter=0xd970, setAudioFilter=0xd97c,
         setDemodMode=0xd798, allocate=0x10cd4)


def engine(agc_mode, demod, audio_filter=6):
    """the object after INO:117-139, with the given AGC mode (0 off ... 3 slow), demodulator (engine numbering: 0 LSB,
    1 USB, ...) and audio filter (6 = audio2700)"""
    import make_firmware_kat as M
    from firmware_image import Image
    r = M.Ref(Image())
    r.call_addr(E["ctor"], SDR)
    r.call_addr(E["enableAGC"], SDR)
    r.call_addr(E["setAGCmode"], SDR, agc_mode)
    r.call_addr(E["disableALSfilter"], SDR)
    r.call_addr(E["disableNoiseBlanker"], SDR)
    r.cpu.call(E["setInputGain"], [SDR], sargs=[1.0])
    r.cpu.call(E["setOutputGain"], [SDR], sargs=[0.5])
    r.cpu.call(E["setIQgainBalance"], [SDR], sargs=[1.02])
    r.call_addr(E["enableAudioFilter"], SDR)
    r.call_addr(E["setAudioFilter"], SDR, audio_filter)
    r.call_addr(E["setDemodMode"], SDR, demod)
    offset = r.cpu.fs(0)
    bl = M.Blocks(r, 32)
    cur, sent = {}, {}
    r.cpu.hooks[M.A["receiveReadOnly"]] = lambda c: c.r.__setitem__(0, cur[c.r[1]])
    r.cpu.hooks[M.A["receiveWritable"]] = lambda c: c.r.__setitem__(0, cur[c.r[1]])
    r.cpu.hooks[M.A["release"]] = lambda c: None
    r.cpu.hooks[E["allocate"]] = lambda c: c.r.__setitem__(0, bl.new())
    r.cpu.hooks[M.A["transmit"]] = lambda c: sent.__setitem__(c.r[2], r.get(c.r[1] + 4, 128, np.int16))

    def run(i, q):
        out = []
        for b in range(len(i) // 128):
            cur[0], cur[1] = bl.new(i[b * 128:(b + 1) * 128]), bl.new(q[b * 128:(b + 1) * 128])
            sent.clear()
            r.call_addr(E["update"], SDR)
            out.append(sent[0].copy())
        return np.concatenate(out)
    return run, offset


def tone(f_hz, amp, n):
    k = np.arange(n)
    a = np.broadcast_to(np.asarray(amp, np.float64), (n,))
    return (a * np.cos(2 * np.pi * f_hz / FS * k) * 32768).astype(np.int16), (a * np.sin(2 * np.pi * f_hz / FS * k) * 32768).astype(np.int16)


def block_rms(x):
    return np.sqrt((x.reshape(-1, 128).astype(np.float64) ** 2).mean(1))


def job_agc(mode):
    run, off = engine(mode, 0)
    nb = (30, 40, 60 if mode == 0 else 280)
    amp = np.concatenate([np.full(nb[0] * 128, 0.01), np.full(nb[1] * 128, 0.3), np.full(nb[2] * 128, 0.01)])
    i, q = tone(off - 1000.0, amp, len(amp))                 # LSB: the side band is below the carrier
    return ("agc_rms_mode%d" % mode, block_rms(run(i, q)))


def job_gain(args):
    demod, audio_hz = args
    run, off = engine(0, demod)
    f = off - audio_hz if demod == 0 else off + audio_hz     # inside the side band for positive audio_hz
    i, q = tone(f, 0.05, 40 * 128)
    y = run(i, q)[24 * 128:]
    return ("gain_demod%d_%+d" % (demod, int(audio_hz)), np.sqrt((y.astype(np.float64) ** 2).mean()) / (0.05 * 32768 / np.sqrt(2)))


def main():
    from make_firmware_tables import HEX
    if not os.path.exists(HEX):
        sys.exit("the reference tree is not here: this script runs in the build container only")
    audio = [100.0, 150.0, 300.0, 1000.0, 2000.0, 2600.0, 2800.0, 3200.0, 4000.0]
    jobs_gain = [(0, a) for a in audio] + [(1, 1000.0), (0, -1000.0), (1, -1000.0)]   # the last two: a tone in the OTHER side band
    with Pool(6) as p:
        res = dict(p.map(job_agc, [0, 1, 2, 3]) + p.map(job_gain, jobs_gain))
    res["audio_hz"] = np.array(audio)
    res["lsb_gain_vs_audio_hz"] = np.array([res.pop("gain_demod0_%+d" % int(a)) for a in audio])
    np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in res.items()})
    print("wrote", OUT)
    for k in sorted(res):
        v = np.asarray(res[k])
        print(k, np.round(v[::10], 1) if v.ndim and len(v) > 20 else np.round(v, 4))


if __name__ == "__main__":
    main()
