"""receiver groups -- per-group retune / PBT / mode tables with
double-buffered masks.  The CPU part checks the host logic (PBT stepping and the
tuningMode table against the oracle's restatement of CTL:330-423,569-612); the GPU
part runs channels of different groups against one oracle chain per channel, with
retunes issued between calls while earlier calls are still queued.

Tolerance: TOL = 1e-5 normwise per channel (feed-forward chains), as in
test_gpu_parity.py; channels of untouched groups must be bit-identical to a run
without any retune.
"""
import numpy as np
import pytest

from cases import K1, TOL


# ---- host logic (no GPU) -------------------------------------------------------------
def test_pbt_step_matches_reference_walk(rdsp, oracle):
    from radiodsp_sdr_rx_amd.chain import pbt_step
    lo, hi = 300.0, 4000.0          # dFLoCut/dFHiCut at boot, GEN:76-77
    olo, ohi = lo, hi
    rng = np.random.default_rng(3)
    for _ in range(400):
        edge, d = int(rng.integers(0, 2)), int(rng.choice([-1, 1]))
        lo, hi = pbt_step(lo, hi, edge, d)
        olo, ohi = oracle.pbt_step(olo, ohi, edge, d)
        assert (lo, hi) == (olo, ohi)
        assert 0.0 <= lo <= 700.0 and 800.0 <= hi <= 4000.0
    # the reference's comparisons: LOCUT stops at 50 on the way down (strict >), reaches 700 on the way up
    lo, hi = 100.0, 900.0
    lo, hi = pbt_step(lo, hi, 0, -1); assert lo == 50.0
    lo, hi = pbt_step(lo, hi, 0, -1); assert lo == 50.0          # (50-50) > 0 is false, CTL:595
    hi = pbt_step(lo, 850.0, 1, -1)[1]; assert hi == 850.0       # (850-50) > 800 is false, CTL:604
    assert pbt_step(650.0, 900.0, 0, +1)[0] == 700.0 and pbt_step(700.0, 900.0, 0, +1)[0] == 700.0
    assert pbt_step(300.0, 3950.0, 1, +1)[1] == 4000.0 and pbt_step(300.0, 4000.0, 1, +1)[1] == 4000.0


def test_pbt_step_rejects_bad_arguments(rdsp):
    import ctypes as C
    lib = rdsp.load()
    a, b = C.c_double(300.0), C.c_double(2700.0)
    assert lib.rdsp_pbt_step(C.byref(a), C.byref(b), 2, 1) != 0
    assert lib.rdsp_pbt_step(C.byref(a), C.byref(b), 0, 0) != 0
    assert lib.rdsp_pbt_step(None, C.byref(b), 0, 1) != 0
    assert (a.value, b.value) == (300.0, 2700.0)


# ---- GPU parity --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _oracle_channel(oracle, base, settings, iq_parts, retunes=None):
    """one oracle chain; settings = dict(demod, lo, hi, nco); retunes[k] applied before part k"""
    cfg = dict(base, demod=settings["demod"], flo_hz=settings["lo"], fhi_hz=settings["hi"], nco_hz=settings["nco"])
    oc = oracle.OracleChain(**cfg)
    outs = []
    for k, part in enumerate(iq_parts):
        for fn in (retunes or {}).get(k, []):
            fn(oc)
        outs.append(oc.process(part)[1])
    return np.concatenate(outs), oc


@pytest.mark.gpu
def test_groups_static_different_filters_modes_offsets(rdsp, oracle, torch_cuda):
    """Three groups with their own pass band, demodulator and tuning offset in one launch."""
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    D = rdsp.DEMOD
    nch, nblk = 7, 32
    iq = synth_iq(nch, nblk * 128)
    group_of = np.array([0, 1, 2, 1, 0, 2, 1], np.uint16)
    settings = [dict(demod="USB", lo=300.0, hi=2700.0, nco=12000.0),
                dict(demod="LSB", lo=-2400.0, hi=-200.0, nco=14600.0),
                dict(demod="AM", lo=-3900.0, hi=3900.0, nco=12900.0)]
    base = dict(fft_l=512, agc_mode="medium", output_gain=0.5)
    ch = Chain(nch, max_blocks_per_call=nblk, **dict(base, demod="USB"))
    ch.set_groups(group_of)
    assert ch.n_groups == 3
    for g, s in enumerate(settings):
        ch.group_setDemodMode(g, D[s["demod"]])
        ch.group_reInitializeFilter(g, s["lo"], s["hi"])
        ch.group_setTuningOffsetHz(g, s["nco"])
    got = ch.process(torch.from_numpy(iq).cuda(), want_f32=True)[1].cpu().numpy()
    for c in range(nch):
        ref, oc = _oracle_channel(oracle, base, settings[group_of[c]], [iq[c]])
        err = np.abs(got[c] - ref).max() / np.abs(ref).max()
        assert err <= TOL, f"channel {c} (group {group_of[c]}): {err:.2e}"
        assert np.abs(ch.group_mask(int(group_of[c])) - oc.mask()).max() < 2e-6


@pytest.mark.gpu
def test_hot_retune_of_one_group_while_calls_are_queued(rdsp, oracle, torch_cuda):
    """PBT on group 1 between calls, no host synchronisation anywhere: the calls queued
    before a retune keep the old mask, the next call uses the new one, other groups are
    bit-identical to a run without retunes.  Several retunes in a row exercise both mask
    buffers and the restage-before-commit path."""
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq, pbt_step
    nch, calls, per = 6, 6, 8
    iq = synth_iq(nch, calls * per * 128)
    parts = [np.ascontiguousarray(iq[:, k * per * 128:(k + 1) * per * 128]) for k in range(calls)]
    dev = [torch.from_numpy(p).cuda() for p in parts]
    group_of = np.array([0, 1, 1, 0, 2, 1], np.uint16)
    base = dict(K1, fft_l=256)

    def run(with_retunes):
        ch = Chain(nch, max_blocks_per_call=per, **base)
        ch.set_groups(group_of)
        ch.group_reInitializeFilter(2, 200.0, 3100.0)
        outs = []
        for k in range(calls):
            if with_retunes:
                if k == 1:
                    ch.group_pbt(1, 1, -1)                   # HICUT 2700 -> 2650
                if k == 2:
                    ch.group_pbt(1, 0, +1)                   # LOCUT 300 -> 350
                    ch.group_pbt(1, 0, +1)                   # restaged before any launch: 400
                if k == 4:
                    ch.group_reInitializeFilter(1, 500.0, 1500.0)
                    ch.group_setTuningOffsetHz(1, 12100.0)
            outs.append(ch.process(dev[k], want_f32=True)[1])
        torch.cuda.synchronize()
        return np.concatenate([o.cpu().numpy() for o in outs], 1), ch

    got, ch = run(True)
    plain, _ = run(False)
    lo, hi = 300.0, 2700.0
    hi1 = pbt_step(lo, hi, 1, -1)[1]
    lo2 = pbt_step(pbt_step(lo, hi1, 0, +1)[0], hi1, 0, +1)[0]
    assert (hi1, lo2) == (2650.0, 400.0)
    retunes = {1: [lambda oc: oc.reinit_filter(300.0, 2650.0)],
               2: [lambda oc: oc.reinit_filter(400.0, 2650.0)],
               4: [lambda oc: oc.reinit_filter(500.0, 1500.0), lambda oc: oc.set_nco_hz(12100.0)]}
    for c in range(nch):
        g = int(group_of[c])
        if g == 1:
            ref, oc = _oracle_channel(oracle, base, dict(demod="USB", lo=300.0, hi=2700.0, nco=12000.0),
                                      [p[c] for p in parts], retunes)
            err = np.abs(got[c] - ref).max() / np.abs(ref).max()
            assert err <= TOL, f"channel {c}: {err:.2e}"
            assert np.abs(ch.group_mask(1) - oc.mask()).max() < 2e-6
            assert np.abs(got[c] - plain[c]).max() > 1e-3        # the retunes did something
        else:
            assert np.array_equal(got[c], plain[c]), f"channel {c} of group {g} changed"


@pytest.mark.gpu
def test_tuning_mode_table_per_group(rdsp, oracle, torch_cuda):
    """tuningMode() (CTL:330-423) per group: filter + demodulator + TuningOffset, checked
    against the oracle's restatement of the table, then through the chain."""
    torch = torch_cuda
    from radiodsp_sdr_rx_amd import RdspError  # noqa: F401
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    names = {v: k for k, v in rdsp.DEMOD.items()}
    nch, nblk = 6, 32
    iq = synth_iq(nch, nblk * 128)
    base = dict(fft_l=512)
    ch = Chain(nch, max_blocks_per_call=nblk, **base)
    ch.set_groups(np.arange(nch, dtype=np.uint16))
    modes = [(0, 14.1e6), (0, 7.03e6), (2, 14.2e6), (3, 3.7e6), (4, 9.5e6), (6, 14.08e6)]
    settings = []
    for g, (mndx, vfo) in enumerate(modes):
        off = ch.group_tuningMode(g, mndx, vfo)
        ok, filt, demod = oracle.tuning_mode(mndx, vfo)
        assert ok and off == oracle.load().orc_demod_tuning_offset(demod)
        lo, hi = oracle.passband(filt, demod)
        ch.group_setTuningOffsetHz(g, float(off))        # what the engine does itself: the carrier from TuningOffset to 0 Hz
        settings.append(dict(demod=names[demod], lo=lo, hi=hi, nco=float(off)))
    assert ch.group_tuningMode(0, 9, 7.1e6) == 0 and b"no menu entry" in rdsp.load().rdsp_last_error()
    ch.group_tuningMode(0, *modes[0])
    ch.group_setTuningOffsetHz(0, settings[0]["nco"])
    got = ch.process(torch.from_numpy(iq).cuda(), want_f32=True)[1].cpu().numpy()
    for c in range(nch):
        ref, _ = _oracle_channel(oracle, base, settings[c], [iq[c]])
        err = np.abs(got[c] - ref).max() / np.abs(ref).max()
        assert err <= TOL, f"group {c} {modes[c]}: {err:.2e}"


@pytest.mark.gpu
def test_group_argument_errors_are_loud(rdsp, torch_cuda):
    from radiodsp_sdr_rx_amd import RdspError
    from radiodsp_sdr_rx_amd.chain import Chain
    ch = Chain(4, max_blocks_per_call=8, **K1)
    with pytest.raises(RdspError):
        ch.group_reInitializeFilter(1, 300.0, 2700.0)          # only group 0 exists
    with pytest.raises(ValueError):
        ch.set_groups(np.zeros(3, np.uint16))
    ch.set_groups(np.array([0, 1, 0, 1], np.uint16))
    with pytest.raises(RdspError):
        ch.group_pbt(2, 0, 1)
    ch.set_groups(None)
    assert ch.n_groups == 1


@pytest.mark.gpu
def test_back_to_back_retunes_while_the_device_is_calls_behind(rdsp, torch_cuda):
    """Two retunes of the same group one call apart, issued while the device is still working on
    earlier calls: the first retune's upload is queued behind those calls when the second retune
    arrives, and must still carry the first mask (its pinned image is not the second one's).  Found by
    the randomised sessions: the call between the two retunes ran with the second mask.  The reference
    run synchronises before every retune; both must give the same bits."""
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    nch, per, calls = 1024, 32, 8
    iq = torch.from_numpy(synth_iq(nch, per * 128)).cuda()       # every call gets the same block of input
    group_of = (np.arange(nch) % 3).astype(np.uint16)
    bands = [(300.0, 2100.0), (500.0, 1000.0), (100.0, 3400.0), (300.0, 2700.0), (600.0, 2400.0)]

    def run(sync_before_retune):
        ch = Chain(nch, max_blocks_per_call=per, **K1)
        ch.set_groups(group_of)
        outs = []
        for k in range(calls):
            if k >= 3:                                # the device is three calls behind by now
                if sync_before_retune:
                    torch.cuda.synchronize()
                ch.group_reInitializeFilter(1, *bands[k - 3])
            outs.append(ch.process(iq))
        torch.cuda.synchronize()
        return np.stack([o.cpu().numpy() for o in outs])

    ref = run(True)
    for _ in range(3):
        got = run(False)
        assert np.array_equal(got, ref)
    assert not np.array_equal(ref[4][1], ref[5][1])      # the retunes do change group 1 from call to call
