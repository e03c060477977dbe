import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib, np_model
from cases import K1, K3
from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
from parity_util import normwise
nch, nblk = 5, 64
iq = synth_iq(nch, nblk * 128)
for name, cfg in (("k2_256", K1), ("usb_512", dict(K1, fft_l=512)), ("als_notch", dict(fft_l=256, demod="USB", als_mode="notch", als_strength=20))):
    f64 = np.stack([np_model.Model(**cfg).process(iq[c]) for c in range(nch)])
    r32 = np.stack([oracle_lib.OracleChain(**cfg).process(iq[c])[1] for c in range(nch)])
    for fir, lab in ((None, "fold"), (2, "two-stage fd"), (0, "direct")):
        ch = Chain(nch, max_blocks_per_call=nblk, **cfg)
        if fir is not None: ch.set_fir_variant(fir)
        o = ch.process(torch.from_numpy(iq).cuda(), want_f32=True)[1].cpu().numpy()
        torch.cuda.synchronize()
        den = np.abs(f64).max(axis=(1, 2))
        eg = np.abs(o - f64).max(axis=(1, 2)) / den
        eo = np.abs(r32 - f64).max(axis=(1, 2)) / den
        print(f"{name:10s} {lab:13s} kernel {ch.front_kernel_name():24s} gpu-f64 {eg.max():.2e} (per ch {np.array2string(eg, precision=1)})  oracle-f64 {eo.max():.2e}  gpu-oracle {normwise(o, r32):.2e}")
