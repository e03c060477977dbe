"""The shapes bench.py times, checked for correctness (the driver's number rests on them).

bench.py's step is ONE call of 512 input blocks (65 536 IQ samples) per channel -- 4096 channels for K2 / K3
(1 GiB of int16 IQ), 8192 for K4 / K5 (2 GiB) -- through a pipelined chain with stage A3 in the throughput form
(rdsp_chain_set_fir_variant 2: 36 decimator frames of 448 outputs and a partial 37th, 256 overlap-save hops at
FFT_L 512, 512 NLMS blocks), the same resident input step after step.  Here: the same chain objects and calls, two
consecutive steps, and sampled channels -- the first, the last, the pair either side of the 4096-channel sub-batch
boundary and of every 64-channel wave boundary nearby, and a random draw -- against the CPU oracle run over the
same two steps as one stream.  The `K3_default` / `K2_default` legs of the bench line (the library's default decimator:
one granule per wave-wide frame beside the tail stage, the row form without one) likewise, and K4's shape under the default.
"""
import numpy as np
import pytest

from cases import K1, K3, K4, TOL
from parity_util import assert_truth_anchored, check_i16, model_run, normwise, oracle_run

pytestmark = pytest.mark.gpu

SHAPES = {
    # name: (chain config, channels, cw input, fir_variant, recursive stage in the chain)
    "K2": (K1, 4096, False, 2, False),
    "K2_default": (K1, 4096, False, None, False),      # no tail stage: the default runs the row form (rdsp_front_rd_kernel)
    "K4_default": (K4, 8192, True, None, False),
    "K3": (K3, 4096, False, 2, True),
    "K3_default": (K3, 4096, False, None, True),
    "K4": (K4, 8192, True, 2, False),
    "K5": (K3, 8192, False, 2, True),
}
BLOCKS = 512
STEPS = 2


@pytest.mark.parametrize("name", list(SHAPES))
def test_bench_step_shape_matches_the_oracle(rdsp, oracle, name):
    import torch
    assert torch.cuda.is_available()
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    cfg, nch, cw, fir, recursive = SHAPES[name]
    n = BLOCKS * 128
    iq = synth_iq(nch, n, cw=cw, n_threads=16)
    dev = torch.from_numpy(iq).cuda()
    ch = Chain(nch, max_blocks_per_call=BLOCKS, fir_variant=fir, **cfg)      # exactly bench.py's chain
    ch.set_pipelined(True)
    o16 = [torch.zeros((nch, n // 4, 2), dtype=torch.int16, device="cuda") for _ in range(STEPS)]
    o32 = [torch.zeros((nch, n // 4, 2), dtype=torch.float32, device="cuda") for _ in range(STEPS)]
    for k in range(STEPS):
        ch.process(dev, out=o16[k], out_f32=o32[k])
    ch.flush()
    torch.cuda.synchronize()
    assert ch.front_kernel_name() == ("rdsp_front_rd_kernel" if name in ("K2_default", "K4_default") else "rdsp_front_fd_kernel")
    rng = np.random.default_rng(len(name))
    edges = [0, 1, 63, 64, 127, 128, 2047, 2048, 4031, 4032, 4095, nch - 65, nch - 64, nch - 2, nch - 1]
    if nch > 4096:
        edges += [4096, 4097, 4159, 4160]                                     # the second sub-batch of the pipelined call
    sample = sorted(set(edges) | set(int(c) for c in rng.integers(0, nch, 24)))
    got16 = np.concatenate([o[sample].cpu().numpy() for o in o16], axis=1)
    got32 = np.concatenate([o[sample].cpu().numpy() for o in o32], axis=1)
    stream = np.concatenate([iq[sample]] * STEPS, axis=1)                     # the two steps as the one stream they are
    r16, r32 = oracle_run(oracle, stream, cfg)
    err = normwise(got32, r32)
    print(f"{name}: {nch} channels x {BLOCKS} blocks x {STEPS} steps, {len(sample)} sampled channels, worst gpu vs oracle {err:.2e}")
    if recursive:
        few = sample[:4] + sample[-2:]                                        # float64 model: a third of a second per channel and step
        idx = [sample.index(c) for c in few]
        assert_truth_anchored(got32[idx], r32[idx], model_run(stream[idx], cfg), name, got16[idx], r16[idx])
        assert err <= TOL                                                     # the metric configuration meets 1e-5 directly
    else:
        assert err <= TOL
        check_i16(got16, r16)
    # every channel of the launch produced finite, non-trivial audio in both steps
    for k in range(STEPS):
        pw = o32[k][..., 0].pow(2).mean(dim=1)
        assert bool(torch.isfinite(pw).all()) and float(pw.min()) > 1e-7
    sc = ch.scalars()
    assert np.isfinite(sc).all()
