"""K5 shape (8192 channels): one chain against two chains of 4096 channels driven alternately
(what launching a call as two channel halves would give).  python tests/micro/split_chain.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import radiodsp_sdr_rx_amd as R
from radiodsp_sdr_rx_amd.chain import Chain, synth_iq

nblk, steps, warm = 512, 100, 30
cfg = dict(R.K_CONFIGS["K5"]["cfg"])
nch = 8192
small = synth_iq(256, nblk * 128, n_threads=8)
iq = torch.from_numpy(np.tile(small, (nch // 256, 1, 1))).cuda()


def run(parts):
    per = nch // parts
    chains = [Chain(per, max_blocks_per_call=nblk, **cfg) for _ in range(parts)]
    for c in chains:
        c.set_pipelined(True)
    ins = [iq[k * per:(k + 1) * per] for k in range(parts)]
    outs = [torch.empty((per, nblk * 32, 2), dtype=torch.int16, device="cuda") for _ in range(parts)]
    for _ in range(warm):
        for c, i, o in zip(chains, ins, outs):
            c.process(i, out=o)
    for c in chains:
        c.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for c, i, o in zip(chains, ins, outs):
            c.process(i, out=o)
    for c in chains:
        c.flush()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{parts} chain(s) of {per} channels: {ms:.3f} ms per 8192-channel step", flush=True)


for parts in (1, 2, 1, 2, 1, 2):
    run(parts)
