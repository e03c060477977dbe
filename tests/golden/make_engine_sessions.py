#!/usr/bin/env python3
"""Random control sessions of the reference's AudioSDR engine, run BY ITS OWN COMPILED CODE (companion of
make_engine_kat.py: same driver, same fixture layout, tests/golden/engine_sessions.npz).

make_engine_kat.py's cases are designed; these are drawn: per session a seeded sequence of the setters the sketch can call
(mode menu, audio filters, AGC modes, ALS on / off / notch / peak, blanker, mute, the three gains), each at a random block,
over a signal whose level and content change -- the kind of input that finds what a designed case did not think of
(a filter re-initialised while its state is live, the lines of the side-band network surviving a stay in AM, the ALS
filter cleared by enableALSfilter but not by disable / enable of the AGC, ...).  tests/test_engine_kat.py holds the CPU
restatement and rdsp_engine_t to the image's audio bit for bit.

Build container only (needs /root/reference); about three minutes.
    python tests/golden/make_engine_sessions.py [--check]
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
OUT = os.path.join(HERE, "engine_sessions.npz")
N_SESSIONS, N_BLOCKS = 6, 48


def session(k):
    from make_engine_kat import tones
    r = np.random.default_rng(1000 + k)
    n = N_BLOCKS * 128
    parts = [(float(r.uniform(4000, 10000)), float(r.uniform(0.02, 0.3))) for _ in range(int(r.integers(1, 5)))] + [(6890.0, float(r.uniform(0.0, 0.3)))]
    env = np.repeat(r.choice([0.02, 0.2, 1.0, 3.0], N_BLOCKS // 4), 4 * 128)
    iq = tones(n, parts, float(r.uniform(0.0, 0.05)), 2000 + k, env=env)
    menu = [("setDemodMode", int(m)) for m in range(7)] + [("setAudioFilter", int(f)) for f in range(11)] + [("setAGCmode", int(a)) for a in range(4)]
    menu += [("enableAGC",), ("enableALSfilter",), ("disableALSfilter",), ("setALSfilterNotch",), ("setALSfilterPeak",), ("setALSfilterAdaptive",),
             ("enableNoiseBlanker",), ("disableNoiseBlanker",), ("setMute", 1), ("setMute", 0), ("enableAudioFilter",)]
    menu += [("setInputGain", float(g)) for g in (0.25, 1.0, 3.0)] + [("setOutputGain", float(g)) for g in (0.2, 0.5, 0.9)] + [("setIQgainBalance", float(g)) for g in (0.95, 1.0, 1.02)]
    calls = []
    for b in sorted(r.integers(0, N_BLOCKS, 14)):
        c = menu[int(r.integers(0, len(menu)))]
        calls.append([int(b)] + list(c))
    return dict(calls=calls, taps=0, iq=iq)


def main():
    from make_firmware_tables import HEX
    if not os.path.exists(HEX):
        sys.exit("the reference tree is not here: this script runs in the build container only")
    import make_engine_kat as K
    t0 = time.time()
    items = [("session%d" % k, session(k)) for k in range(N_SESSIONS)]
    with Pool(6) as p:
        parts = p.map(K.run_case, items, chunksize=1)
    res = {"case_names": np.array([n for n, _ in items])}
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
