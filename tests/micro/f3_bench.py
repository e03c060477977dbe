import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import radiodsp_sdr_rx_amd as R
from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
kc = R.K_CONFIGS["K3"]; nch, nblk = 4096, 512
iq = torch.from_numpy(synth_iq(nch, nblk*128, n_threads=16)).cuda()
out = torch.empty((nch, nblk*32, 2), dtype=torch.int16, device="cuda")
for nb, sw in [(0,0),(1,0),(1,1)]:
    ch = Chain(nch, max_blocks_per_call=nblk, **kc["cfg"]); ch.set_pipelined(True)
    if nb: ch.enableNoiseBlanker()
    ch.swapIQ(bool(sw))
    for _ in range(3): ch.process(iq, out=out)
    ch.flush(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(20): ch.process(iq, out=out)
    ch.flush(); torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/20
    print(f"K3 noise_blanker={nb} swapIQ={sw}: {dt*1e3:.3f} ms/step, {nch*nblk*128/dt/1e9:.1f} Gsamples/s")
# SAM: the PLL stage is serial per sample, one channel per lane (64 waves at 4096 channels)
cfg = dict(kc["cfg"], demod="SAM", flo_hz=-3900.0, fhi_hz=3900.0, als_mode="off")
for piped in (False, True):
    ch = Chain(nch, max_blocks_per_call=nblk, **cfg); ch.set_pipelined(piped)
    for _ in range(2): ch.process(iq, out=out)
    ch.flush(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): ch.process(iq, out=out)
    ch.flush(); torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"K3 front + SAM + AGC, pipelined={piped}: {dt*1e3:.3f} ms/step, {nch*nblk*128/dt/1e9:.1f} Gsamples/s")
# K4 (FFT_L 4096, four waves per channel) with the blanker: frequency-domain decimator (the pre-pass goes
# round the waves) against the direct form such chains used before
kc = R.K_CONFIGS["K4"]; nch4 = 8192
iq4 = torch.from_numpy(synth_iq(nch4, nblk*128, cw=True, n_threads=16)).cuda()
out4 = torch.empty((nch4, nblk*32, 2), dtype=torch.int16, device="cuda")
for nb, fir in [(0, -1), (1, -1), (1, 0)]:
    ch = Chain(nch4, max_blocks_per_call=nblk, **kc["cfg"])
    if nb: ch.enableNoiseBlanker()
    ch.set_fir_variant(fir)
    for _ in range(3): ch.process(iq4, out=out4)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): ch.process(iq4, out=out4)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"K4 noise_blanker={nb} decimator={'frequency domain' if fir else 'direct form'} ({ch.front_kernel_name()}): {dt*1e3:.3f} ms/step, {nch4*nblk*128/dt/1e9:.1f} Gsamples/s")
