#!/usr/bin/env python3
"""The sketch's whole audio path, end to end, computed BY THE REFERENCE'S OWN COMPILED CODE.

INO:71-89 wires   IQ input -> preProcessor -> SDR -> record queues Q_in_L / Q_in_R,   loop() (INO:198) runs
doConvolutionalProcessing between those and the play queues Q_out_L / Q_out_R.  Every piece is in the firmware image
(pre_compiled/RadioDSP_SDR_RX.ino.hex) and runs under tests/golden/thumb_emu.py: AudioSDRpreProcessor::update (0xee88),
AudioSDR::update (0xe730), doConvolutionalProcessing (0x70d0).  This script chains them block by block -- hooks hand the
blocks across, as the audio library's transmit / receive and the queues would -- at the sketch's start-up settings
(INO:117-139,172-183: auto I2S-error detection on, AGC medium, audio2700, LSBmode, gains 1.0 / 0.5 / 1.020, NR 15,
filter 300 ... 4000 Hz) and writes to tests/golden/sketch_kat.npz, per case,
  <case>_iq         int16 [n, 2]        what the codec delivers
  <case>_pre        int16 [n, 2]        what the pre-processor hands on          <case>_pre_state  int16 [blocks, 4]
  <case>_sdr        int16 [n]           what the engine transmits (on both outputs)
  <case>_audio      int16 [n, 2]        what doConvolutionalProcessing plays
plus pre-processor-only cases (a slipped rail found and repaired, swapIQ).  Data only.

Build container only (needs /root/reference); about five minutes.
    python tests/golden/make_sketch_kat.py [--check]
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "sketch_kat.npz")
I16 = np.int16


def slip(iq, rail):
    """the I2S start-up fault: `rail` arrives one sample late"""
    x = iq.copy()
    x[1:, rail] = iq[:-1, rail]
    x[0, rail] = 0
    return x


def inputs():
    from make_engine_kat import tones
    c = {}
    c["sketch_path"] = ("full", tones(96 * 128, [(8390.0 - 600, 0.10), (8390.0 - 1500, 0.07), (8390.0 - 2300, 0.04), (8390.0 + 900, 0.10)], 0.01, 51))
    c["sketch_path_slip"] = ("full", slip(tones(40 * 128, [(8390.0 - 1000, 0.25), (8390.0 - 1700, 0.02)], 0.004, 52), 1))
    c["pre_slip_i"] = ("pre", slip(tones(40 * 128, [(5000.0, 0.3)], 0.003, 53), 0))      # I late: the first remedy (I later still) is wrong, the second right
    c["pre_slip_q"] = ("pre", slip(tones(40 * 128, [(-7000.0, 0.3)], 0.003, 54), 1))
    c["pre_clean"] = ("pre", tones(20 * 128, [(3000.0, 0.2)], 0.01, 55))
    c["pre_noise"] = ("pre", tones(12 * 128, [], 0.1, 56))                                # no carrier: nothing is counted
    c["pre_swap"] = ("pre_swap", tones(6 * 128, [(3000.0, 0.2)], 0.01, 57))
    return c


def run_case(item):
    name, (kind, iq) = item
    from engine_ref import EngineRef, PreRef
    import make_firmware_kat as M
    pre = PreRef()
    if kind == "pre_swap":
        pre.swapIQ(True)
    nb = len(iq) // 128
    po, ps = np.zeros_like(iq), np.zeros((nb, 4), I16)
    for b in range(nb):
        i, q = pre.update(iq[b * 128:(b + 1) * 128, 0], iq[b * 128:(b + 1) * 128, 1])
        po[b * 128:(b + 1) * 128, 0], po[b * 128:(b + 1) * 128, 1] = i, q
        ps[b] = pre.state()
    res = {name + "_iq": iq, name + "_pre": po, name + "_pre_state": ps}
    if kind == "full":
        e = EngineRef()
        sdr = e.run(po[:, 0], po[:, 1])
        s = M.Sketch(e.r.im)
        s.setup()
        o16, _ = s.process(np.stack([sdr, sdr], 1), 15.0)
        res[name + "_sdr"], res[name + "_audio"] = sdr, o16
    return res


def main():
    from make_firmware_tables import HEX
    if not os.path.exists(HEX):
        sys.exit("the reference tree is not here: this script runs in the build container only")
    t0 = time.time()
    with Pool(7) as p:
        parts = p.map(run_case, sorted(inputs().items(), key=lambda kv: -len(kv[1][1])), chunksize=1)
    res = {}
    for part in parts:
        res.update(part)
    if "--check" in sys.argv:
        old = np.load(OUT)
        bad = [k for k in res if k not in old.files or not np.array_equal(np.asarray(res[k]), old[k])] + [k for k in old.files if k not in res]
        print("identical (%d arrays)" % len(res) if not bad else "DIFFERENT: %s" % bad, "%.0f s" % (time.time() - t0))
        sys.exit(1 if bad else 0)
    np.savez_compressed(OUT, **res)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(res), "arrays, %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
