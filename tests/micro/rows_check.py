"""ad hoc: the row forms of the decimator (fir_variant 5 / 6) against the direct form, several FFT_L and call splits"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import radiodsp_sdr_rx_amd as R
from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
R.load()
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cases import K1, K3, K4
CFGS = {"k2": K1, "k3": K3, "u512": dict(fft_l=512, demod="USB", agc_mode="fast"), "u1024": dict(fft_l=1024, demod="USB"),
        "l2048": dict(fft_l=2048, demod="LSB", nco_hz=14600.0, flo_hz=-2700.0, fhi_hz=-300.0), "k4": K4}
def run(iq, cfg, fir, calls):
    nch, n = iq.shape[0], iq.shape[1]
    try:
        ch = Chain(nch, max_blocks_per_call=n // 128 // calls, fir_variant=fir, **cfg)
    except Exception as e:
        return None
    o = []
    step = n // calls
    for k in range(calls):
        a, b = ch.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * step:(k + 1) * step])).cuda(), want_f32=True)
        torch.cuda.synchronize()
        o.append(b.cpu().numpy())
    return np.concatenate(o, 1), ch.front_kernel_name()
for name, cfg in CFGS.items():
    try:
        probe = Chain(2, max_blocks_per_call=512, **cfg)
    except Exception as e:
        print(name, "cfg failed", e); continue
    unit = probe.call_unit_blocks
    nblk = max(96, 6 * unit)
    iq = synth_iq(3, nblk * 128)
    ref, _ = run(iq, cfg, 0, 1)
    for fir in (4, 2, 5, 6):
        for calls in (1, 2, 3):
            if (nblk // calls) % unit: continue
            r = run(iq, cfg, fir, calls)
            err = np.abs(r[0] - ref).max() / np.abs(ref).max()
            print(f"{name:6s} fir {fir} calls {calls} {r[1]:22s} err {err:.2e}", flush=True)
