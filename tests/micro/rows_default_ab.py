"""same-box A/B: the default decimator form against the row form (fir_variant 5), chains without a tail stage"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import radiodsp_sdr_rx_amd as R
from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
R.load()
nch, nblk = 4096, 512
base = synth_iq(8, nblk * 128)
iq = torch.from_numpy(np.ascontiguousarray(np.tile(base, (nch // 8, 1, 1)))).cuda()
for name, cfg in (("256", dict(fft_l=256, demod="USB")), ("512", dict(fft_l=512, demod="USB", agc_mode="fast")),
                  ("512+NR", dict(fft_l=512, demod="USB", spectral_nr=1, spectral_level=2.0, agc_mode="medium")),
                  ("1024", dict(fft_l=1024, demod="USB")), ("2048", dict(fft_l=2048, demod="LSB", flo_hz=-2700.0, fhi_hz=-300.0)),
                  ("4096", dict(fft_l=4096, demod="CW_USB", flo_hz=450.0, fhi_hz=950.0, agc_mode="fast"))):
    for rep in range(2):
        for fir in (-1, 5, 2):
            ch = Chain(nch, max_blocks_per_call=nblk, fir_variant=fir, **cfg)
            for _ in range(5):
                ch.process(iq)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                ch.process(iq)
            torch.cuda.synchronize()
            print(f"FFT_L {name:7s} variant {fir:2d} rep {rep} {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms  {ch.front_kernel_name()}", flush=True)
            del ch
