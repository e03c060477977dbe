#!/usr/bin/env python3
"""Known answers of the reference's AudioSDR engine, computed BY ITS OWN COMPILED CODE.

`AudioSDR SDR;` (INO:54) is the un-vendored engine between the pre-processor and the convolution stage.  Its compiled
update() (ITCM 0xe730 of pre_compiled/RadioDSP_SDR_RX.ino.hex) runs under tests/golden/thumb_emu.py; this script drives
it (tests/golden/engine_ref.py) through the sketch's settings and menus on seeded int16 IQ and writes, per case,
  <case>_iq      int16 [n, 2]   the input blocks
  <case>_calls   JSON           [[block, method, args...], ...]: the engine's setters, called before that block
  <case>_out     int16 [n]      what update() transmitted
  <case>_final   float32 [8]    oscillator phase, AGC gain / envelope / hang counter / active flag, PLL Hz / lock, blanker hit
  <case>_tap_*   float32        for the principal cases, the float buffers of the object after each stage of the first
                                blocks (conversion, blanker, IF filter, mixer, Hilbert pair, demodulator, audio filter, AGC, ALS)
to tests/golden/engine_kat.npz, plus the object's tables as the constructor leaves them (AGC curve, sine table) and
sample points of the image's expf.  Data only.  oracle/rdsp_engine_oracle.c (CPU) and rdsp_engine_t (GPU) are held to
these bits by tests/test_engine_kat.py.

Build container only (needs /root/reference); about twelve minutes on seven cores (the slow AGC's hang is 689 blocks).
    python tests/golden/make_engine_kat.py            writes the fixture
    python tests/golden/make_engine_kat.py --check    recomputes everything and compares with the committed fixture
"""
import json
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "engine_kat.npz")
FS = 44100.0
I16 = np.int16
TAP_NAMES = ["in", "nb", "pre", "mix", "hilbert", "demod", "filt", "agc", "als"]


def q(x):
    return np.clip(np.round(x * 32767.0), -32768, 32767).astype(I16)


def tones(n, parts, noise=0.0, seed=0, env=None):
    """parts: [(Hz, amplitude)], a positive frequency turns counter-clockwise in I + jQ"""
    t = np.arange(n)
    z = np.zeros(n, np.complex128)
    for f, a in parts:
        z += a * np.exp(2j * np.pi * (f / FS) * t + 1j * (0.3 + 0.7 * f / 1000.0))
    if noise:
        r = np.random.default_rng(seed)
        z += noise * (r.standard_normal(n) + 1j * r.standard_normal(n))
    if env is not None:
        z *= env
    return np.stack([q(z.real), q(z.imag)], 1)


def steps(levels):
    return np.concatenate([np.full(nb * 128, a) for a, nb in levels])


# the mode menu's pairs of calls (CTL:330-407) with the numbers the compiled tuningMode() passes
AM, CW, A2100, A2700, A3100 = 0, 1, 3, 6, 8
LSB, USB, CWL, CWU, AMm, SAM = range(6)


def cases():
    c = {}
    off = dict(lsb=8390.0, usb=5390.0, cwl=7390.0, cwu=6390.0, am=6890.0)
    n = 48 * 128
    c["lsb_sketch"] = dict(calls=[], taps=8, iq=tones(n, [(off["lsb"] - 700, 0.12), (off["lsb"] - 1900, 0.08), (off["lsb"] + 1000, 0.2), (-3000, 0.05)], 0.01, 1))
    n = 32 * 128
    c["usb_fast"] = dict(calls=[[0, "setAGCmode", 1], [0, "setAudioFilter", A2700], [0, "setDemodMode", USB]], taps=4,
                         iq=tones(n, [(off["usb"] + 500, 0.1), (off["usb"] + 2300, 0.05), (off["usb"] - 800, 0.1)], 0.01, 2))
    c["cw_lsb_slow"] = dict(calls=[[0, "setAGCmode", 3], [0, "setAudioFilter", CW], [0, "setDemodMode", CWL]], taps=4,
                            iq=tones(n, [(off["cwl"] - 700, 0.2)], 0.02, 3, env=(np.arange(n) // 2646 % 2).astype(float)))
    c["cw_usb_2100"] = dict(calls=[[0, "setAudioFilter", A2100], [0, "setDemodMode", CWU]], taps=0,
                            iq=tones(n, [(off["cwu"] + 700, 0.15), (off["cwu"] - 700, 0.15)], 0.02, 4))
    t = np.arange(n)
    am_env = 1.0 + 0.6 * np.sin(2 * np.pi * 440.0 / FS * t) + 0.3 * np.sin(2 * np.pi * 1300.0 / FS * t)
    c["am"] = dict(calls=[[0, "setAudioFilter", AM], [0, "setDemodMode", AMm]], taps=6, iq=tones(n, [(off["am"], 0.2)], 0.005, 5, env=am_env))
    n = 48 * 128
    t = np.arange(n)
    am_env = 1.0 + 0.5 * np.sin(2 * np.pi * 600.0 / FS * t)
    c["sam"] = dict(calls=[[0, "setAudioFilter", AM], [0, "setDemodMode", SAM]], taps=8, iq=tones(n, [(off["am"] + 3.0, 0.3)], 0.003, 6, env=am_env))
    c["sam_no_carrier"] = dict(calls=[[0, "setAudioFilter", AM], [0, "setDemodMode", SAM]], taps=0, iq=tones(16 * 128, [(500.0, 0.1)], 0.02, 7))
    c["mode6_agc_off"] = dict(calls=[[0, "setAGCmode", 0], [0, "setDemodMode", 6]], taps=3, iq=tones(16 * 128, [(off["usb"] + 900, 0.1), (off["usb"] - 900, 0.1)], 0.01, 8))
    for k in range(11):                                        # every audio filter id setAudioFilter knows, 10 = none
        c["audio_id%d" % k] = dict(calls=[[0, "setAGCmode", 0], [0, "setAudioFilter", k]], taps=0,
                                   iq=tones(12 * 128, [(off["lsb"] - f, 0.04) for f in (100, 300, 700, 1500, 2500, 3300, 4500)], 0.01, 20 + k))
    for name, mode, tail in (("agc_fast", 1, 80), ("agc_medium", 2, 220), ("agc_slow", 3, 720)):   # a 30 dB step up and down: attack, hang, recovery
        lv = steps([(0.01, 20), (0.3, 30), (0.01, tail)])
        c[name] = dict(calls=[[0, "setAGCmode", mode]], taps=0, iq=tones(len(lv), [(off["lsb"] - 1000, 1.0)], 0.0, 0, env=lv))
    c["agc_overdrive"] = dict(calls=[[0, "setAGCmode", 1], [0, "setInputGain", 8.0]], taps=2, iq=tones(12 * 128, [(off["lsb"] - 1000, 0.9)], 0.05, 30))
    c["wrap_agc_off"] = dict(calls=[[0, "setAGCmode", 0], [0, "setInputGain", 6.0], [0, "setOutputGain", 1.0]], taps=0,
                             iq=tones(6 * 128, [(off["lsb"] - 1000, 0.7)], 0.0, 0))        # |audio| x gain > 1: the int16 store wraps
    n = 40 * 128
    dsp = [[0, "enableAGC"], [0, "enableALSfilter"], [0, "setALSfilterNotch"], [0, "setALSfilterAdaptive"]]          # CTL:258-261
    c["als_notch"] = dict(calls=dsp, taps=6, iq=tones(n, [(off["lsb"] - 1000, 0.15), (off["lsb"] - 400, 0.03), (off["lsb"] - 2100, 0.03)], 0.01, 31))
    c["als_peak"] = dict(calls=[[0, "enableALSfilter"], [0, "setALSfilterPeak"], [0, "setALSfilterAdaptive"]], taps=0,
                         iq=tones(24 * 128, [(off["lsb"] - 800, 0.1)], 0.03, 32))
    c["als_on_off_on"] = dict(calls=dsp + [[10, "disableALSfilter"], [16, "enableALSfilter"]], taps=0, iq=tones(24 * 128, [(off["lsb"] - 1200, 0.1)], 0.01, 33))
    x = tones(24 * 128, [(off["lsb"] - 900, 0.05)], 0.005, 34)
    r = np.random.default_rng(35)
    for p in r.integers(300, 24 * 128 - 300, 14):
        x[p:p + int(r.integers(1, 4))] = r.choice([-30000, 30000], 2)
    c["blanker_on"] = dict(calls=[[0, "enableNoiseBlanker"]], taps=8, iq=x)
    c["mute"] = dict(calls=[[2, "setMute", 1], [4, "setMute", 0]], taps=0, iq=tones(8 * 128, [(off["lsb"] - 900, 0.1)], 0.01, 36))
    menu = [[0, "setAudioFilter", A2700], [0, "setDemodMode", USB], [8, "setAudioFilter", AM], [8, "setDemodMode", AMm], [16, "setAudioFilter", AM],
            [16, "setDemodMode", SAM], [28, "setAudioFilter", CW], [28, "setDemodMode", CWU], [34, "setAudioFilter", A2700], [34, "setDemodMode", LSB],
            [37, "setAGCmode", 0], [39, "setAGCmode", 3]]
    c["menu_walk"] = dict(calls=menu, taps=0, iq=tones(44 * 128, [(6890.0, 0.2), (6890.0 + 1000, 0.08), (6890.0 - 1400, 0.08)], 0.01, 37))
    c["gains"] = dict(calls=[[0, "setInputGain", 0.5], [0, "setOutputGain", 0.9], [0, "setIQgainBalance", 0.97], [3, "setInputGain", 12.0], [5, "setInputGain", -1.0]],
                      taps=0, iq=tones(8 * 128, [(off["lsb"] - 900, 0.05)], 0.01, 38))
    z = np.zeros((4 * 128, 2), I16)
    c["silence_then_signal"] = dict(calls=[], taps=0, iq=np.concatenate([z, tones(8 * 128, [(off["lsb"] - 900, 0.1)], 0.0, 0), z]))
    r = np.random.default_rng(39)
    c["rails"] = dict(calls=[], taps=0, iq=np.where(r.random((8 * 128, 2)) < 0.5, 32767, -32768).astype(I16))
    return c


def run_case(item):
    name, case = item
    from engine_ref import EngineRef
    e = EngineRef(taps=case["taps"] > 0)
    iq, calls = case["iq"], case["calls"]
    nb = len(iq) // 128
    out = np.zeros(nb * 128, I16)
    for b in range(nb):
        for call in calls:
            if call[0] == b:
                if call[1] == "setALSfilterPeak":
                    e.poke_u8(3549, 0)                       # the image has no such setter (the sketch never calls it): the flag itself
                elif call[1] == "enableNoiseBlanker":
                    e.poke_u8(0x2258, 1)                     # likewise: the constructor's own default
                else:
                    e.call(call[1], *call[2:])
        out[b * 128:(b + 1) * 128] = e.update(iq[b * 128:(b + 1) * 128, 0], iq[b * 128:(b + 1) * 128, 1])
        if e.taps is not None and b + 1 == case["taps"]:
            e.r.cpu.watch.clear()
    res = {name + "_iq": iq, name + "_calls": np.array(json.dumps(calls)), name + "_out": out}
    res[name + "_final"] = np.array([e.glob("nco_phase")[0], e.obj(0xe08, 1, np.float32)[0], e.obj(0x1030, 1, np.float32)[0], float(e.obj(0x1038, 1, np.int32)[0]),
                                     float(e.obj(0x103c, 1, np.uint8)[0]), e.obj(0x227c, 1, np.float32)[0], float(e.obj(0x2281, 1, np.uint8)[0]),
                                     float(e.obj(0x2259, 1, np.uint8)[0])], np.float32)
    if e.taps is not None:
        for k in TAP_NAMES:
            if k in e.taps:
                res[name + "_tap_" + k] = np.stack(e.taps[k][:case["taps"]])
    return res


def tables():
    from engine_ref import EngineRef, SDR
    e = EngineRef(sketch_setup=False)
    res = {"agc_curve": e.obj(0xe10, 130, np.float32), "sine257": e.obj(0x285c, 257, np.float32), "sam_gains": e.obj(0x2294, 4, np.float32)}
    xs = np.concatenate([np.linspace(-12.0, 1.0, 257), np.array([0.0, 1e-10, -1e-10, 0.3, -0.3, 0.4, 88.0, -100.0])]).astype(np.float32)
    ys = []
    for x in xs:
        e.r.cpu.call(0x13690, [], sargs=[float(x)])
        ys.append(e.r.cpu.fs(0))
    res["expf_x"], res["expf_y"] = xs, np.array(ys, np.float32)
    offs = []
    for mode in range(7):
        e.call("setDemodMode", mode)
        offs.append(e.tuning_offset)
    res["tuning_offsets"] = np.array(offs, np.float32)
    return res


def main():
    from make_firmware_tables import HEX
    if not os.path.exists(HEX):
        sys.exit("the reference tree is not here: this script runs in the build container only")
    t0 = time.time()
    cs = cases()
    order = sorted(cs.items(), key=lambda kv: -len(kv[1]["iq"]))
    with Pool(7) as p:
        parts = p.map(run_case, order, chunksize=1)
    res = tables()
    for part in parts:
        res.update(part)
    res["case_names"] = np.array(sorted(cs))
    if "--check" in sys.argv:
        old = np.load(OUT)
        bad = [k for k in res if k not in old.files or not np.array_equal(np.asarray(res[k]), old[k])] + [k for k in old.files if k not in res]
        print("identical (%d arrays)" % len(res) if not bad else "DIFFERENT: %s" % bad, "%.0f s" % (time.time() - t0))
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(res), "arrays, %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
