"""The frequency-domain decimator on 16-lane rows (rdsp_chain_set_fir_variant 5, rdsp_front_rd_kernel): parity and
split invariance on the GPU.  A module of its own: the form is chosen explicitly, so the module-wide `front_form`
matrix of test_gpu_parity.py would only repeat it."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from cases import K1, K3, K4, TOL  # noqa: E402
from parity_util import normwise  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


@pytest.mark.parametrize("nco_hz", [12000.0, 12345.678])
@pytest.mark.parametrize("name,cfg", [("k2", K1), ("k3", K3), ("usb_1024", dict(fft_l=1024, demod="USB", agc_mode="medium")),
                                      ("lsb_2048_agc", dict(fft_l=2048, demod="LSB", agc_mode="fast", flo_hz=-2700.0, fhi_hz=-300.0)),
                                      ("k4", K4)])
def test_row_form_decimator_meets_tolerance_and_is_split_invariant(rdsp, oracle, torch_cuda, name, cfg, nco_hz):
    """rdsp_chain_set_fir_variant(5): the frequency-domain decimator on 16-lane rows (rdsp_front_rd_kernel: 256-point
    windows, four per wave, two frames per granule).  The same exact linear convolution -- TOL against the direct
    form, one call -- and the same bits for any call split, with a retune / gain / swap / balance session at fixed
    stream positions (a frame's arithmetic may not depend on the row or pass it lands in, nor on whether its first
    64 quads came in with this call or the one before).  With the noise blanker on the chain runs the one-granule
    form of rdsp_front_fd_kernel, split-invariant too."""
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    cfg = dict(cfg, nco_hz=nco_hz)
    nch = 3
    gran = Chain(nch, max_blocks_per_call=64, **cfg).call_unit_blocks
    n_gran = 16 if gran <= 16 else 8
    nblk = n_gran * gran
    iq = synth_iq(nch, nblk * 128, cw=(name == "k4"))
    iq[:, 5000:5003] = 30000
    script = {n_gran // 4: lambda ch: ch.setTuningOffsetHz(nco_hz - 2468.3),
              n_gran // 2: lambda ch: (ch.setInputGain(0.7), ch.swapIQ(True)),
              3 * n_gran // 4: lambda ch: (ch.setIQgainBalance(1.02), ch.swapIQ(False)),
              7 * n_gran // 8: lambda ch: (ch.setIQgainBalance(1.0), ch.setInputGain(1.3))}

    def run(cuts, fir, blanker=False, session=True):
        ch = Chain(nch, max_blocks_per_call=nblk, fir_variant=fir, **cfg)
        if blanker:
            ch.enableNoiseBlanker()
        o, f = [], []
        sc = script if session else {}
        edges = sorted(set(cuts) | set(sc))
        for a, b in zip([0] + edges, edges + [n_gran]):
            if a in sc:
                sc[a](ch)
            x16, x32 = ch.process(torch.from_numpy(np.ascontiguousarray(iq[:, a * gran * 128:b * gran * 128])).cuda(), want_f32=True)
            torch.cuda.synchronize()
            o.append(x16.cpu().numpy())
            f.append(x32.cpu().numpy())
        return np.concatenate(o, 1), np.concatenate(f, 1), ch.scalars(), ch.front_kernel_name()

    _, plain, _, kn = run([], 5, session=False)
    assert kn == "rdsp_front_rd_kernel"
    _, direct, _, _ = run([], 0, session=False)
    assert np.abs(direct).max() > 0.01 and normwise(plain, direct) <= TOL
    one, one32, sc, _ = run([], 5)
    _, d32, _, _ = run([], 0)
    assert normwise(one32, d32) <= TOL
    rng = np.random.default_rng(31)
    for trial in range(4):
        cuts = sorted(set(int(x) for x in rng.integers(1, n_gran, size=rng.integers(1, 6))))
        o, f, s2, _ = run(cuts, 5)
        assert np.array_equal(o, one) and np.array_equal(f, one32) and np.array_equal(s2, sc), (name, cuts)
    o, f, s2, _ = run(list(range(1, n_gran)), 5)      # every call unit its own call
    assert np.array_equal(o, one) and np.array_equal(f, one32) and np.array_equal(s2, sc)
    b16, b32, bs, kn = run([], 5, blanker=True)
    assert kn == "rdsp_front_fd_kernel"
    o, f, s2, _ = run([n_gran // 3, n_gran // 2 + 1], 5, blanker=True)
    assert np.array_equal(o, b16) and np.array_equal(f, b32) and np.array_equal(s2, bs)
    d16, dd32, _, _ = run([], -1, blanker=True)
    assert np.array_equal(b16, d16) and np.array_equal(b32, dd32)   # the same kernel as the default then
