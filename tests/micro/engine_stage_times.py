"""Where rdsp_engine_update's time goes: 4096 receivers x 32 blocks, stages switched off one at a time (same box).
usage (GPU box): python tests/micro/engine_stage_times.py [other/librdsp_hip.so]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_lib  # noqa: E402  (the tables only)
if len(sys.argv) > 1:   # another build of the library (same-box A/B)
    from radiodsp_sdr_rx_amd import _lib
    _lib.use_library(sys.argv[1])
from radiodsp_sdr_rx_amd.chain import synth_iq  # noqa: E402
from radiodsp_sdr_rx_amd.engine import Engine  # noqa: E402

nch, nblk = 4096, 32
iq = torch.from_numpy(synth_iq(nch, nblk * 128, n_threads=8)).cuda()
out = torch.empty_like(iq)


def run(label, setup):
    e = Engine(nch, max_blocks_per_call=nblk, tables=oracle_lib.engine_tables())
    e.sketch_setup()
    setup(e)
    for _ in range(5):
        e.update(iq, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        e.update(iq, out=out)
    torch.cuda.synchronize()
    print(f"{label:34s} {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per call")


run("sketch defaults (LSB, 2700, AGC)", lambda e: None)
run("AGC off", lambda e: e.setAGCmode(0))
run("audio filter off", lambda e: e.setAudioFilter(10))
run("AGC and audio filter off", lambda e: (e.setAGCmode(0), e.setAudioFilter(10)))
run("muted (pack only differs)", lambda e: e.setMute(1))
run("ALS on", lambda e: (e.enableALSfilter(), e.setALSfilterNotch(), e.setALSfilterAdaptive()))
run("blanker on", lambda e: e.enableNoiseBlanker())
run("AM", lambda e: (e.setAudioFilter(0), e.setDemodMode(4)))
run("SAM", lambda e: (e.setAudioFilter(0), e.setDemodMode(5)))
