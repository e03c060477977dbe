"""Randomised control sessions: every engine setter of the boundary (modes, filters, PBT, tuning
offsets, AGC, ALS / DSP-NR / spectral NR, noise blanker, IQ swap, gains, mute, the IIR filter bank)
fired in random order between calls of random length, the same script on two chains:
one plain (everything on the caller's stream), one pipelined with channel sub-batches (tail stage,
SAM and biquad stages on the internal stream, three rotating intermediate buffers, group records
rewritten in stream order).  Same call split, same kernels: the int16 audio, the per-channel
scalars and both NLMS coefficient sets must be bit-identical -- any missing stream dependency
between a setter, a stage and the next call shows up as a difference.

A third chain replays the script on a different channel partition (the first 63 channels only) and
must reproduce those channels bit for bit: nothing may depend on which other channels share a launch.
"""
import numpy as np
import pytest

from cases import K3

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("front_form")]   # every test under both front kernels (conftest.py)

NCH, MAXBLK, NGROUPS = 150, 32, 3


def make_script(rng, n_ops=40, granule=8):
    ops = []
    for _ in range(n_ops):
        kind = rng.choice(["proc", "proc", "proc", "als", "nr", "spec", "agc", "gdemod", "gfilt", "gpbt", "goff", "gaf",
                           "nb", "swap", "slip", "mute", "ogain", "igain", "bal", "afk", "specform", "emode"])
        g = int(rng.integers(0, NGROUPS))
        if kind == "proc":
            ops.append(("proc", granule * int(rng.integers(1, 32 // granule + 1))))
        elif kind == "als":
            ops.append(("als", str(rng.choice(["off", "notch", "peak"]))))
        elif kind == "nr":
            ops.append(("nr", int(rng.choice([0, 0, 20, 30]))))
        elif kind == "spec":
            ops.append(("spec", int(rng.choice([0, 1, 1, 2])), float(rng.choice([1.0, 2.0, 3.0]))))
        elif kind == "agc":
            ops.append(("agc", str(rng.choice(["off", "fast", "medium", "slow"]))))
        elif kind == "gdemod":
            ops.append(("gdemod", g, str(rng.choice(["USB", "LSB", "CW_USB", "CW_LSB", "AM", "SAM"]))))
        elif kind == "gfilt":
            lo = float(rng.choice([100.0, 300.0, 500.0]))
            ops.append(("gfilt", g, lo, lo + float(rng.choice([500.0, 1800.0, 2400.0, 3300.0]))))
        elif kind == "gpbt":
            ops.append(("gpbt", g, int(rng.integers(0, 2)), int(rng.choice([-1, 1]))))
        elif kind == "goff":
            ops.append(("goff", g, float(rng.choice([12000.0, 11300.0, 12700.0, 14600.0]))))
        elif kind == "gaf":
            ops.append(("gaf", g, int(rng.integers(0, 6))))
        elif kind == "nb":
            ops.append(("nb", bool(rng.integers(0, 2)), float(rng.choice([6.0, 10.0, 20.0]))))
        elif kind == "swap":
            ops.append(("swap", bool(rng.integers(0, 2))))
        elif kind == "slip":
            ops.append(("slip", int(rng.choice([-1, 0, 0, 1]))))
        elif kind == "mute":
            ops.append(("mute", bool(rng.integers(0, 4) == 0)))
        elif kind == "ogain":
            ops.append(("ogain", float(rng.choice([0.25, 0.5, 1.0]))))
        elif kind == "igain":
            ops.append(("igain", float(rng.choice([0.5, 1.0, 1.5]))))
        elif kind == "bal":
            ops.append(("bal", float(rng.choice([1.0, 1.02, 0.97]))))
        elif kind == "afk":
            ops.append(("afk", int(rng.integers(0, 2))))
        elif kind == "specform":     # SPEC:229-232 as written / the exact-arithmetic equivalent
            ops.append(("specform", bool(rng.integers(0, 2))))
        elif kind == "emode":        # NR:73's running energy / the per-block re-start
            ops.append(("emode", bool(rng.integers(0, 2))))
    ops.append(("proc", 2 * granule if granule <= 16 else granule))
    ops.append(("proc", granule))
    return ops


def make_chain(n_channels, pipelined, sub_batch, cfg=K3):
    from radiodsp_sdr_rx_amd.chain import Chain
    ch = Chain(n_channels, max_blocks_per_call=MAXBLK, **cfg)
    ch.set_groups((np.arange(n_channels) % NGROUPS).astype(np.uint16))
    ch.set_pipelined(pipelined)
    if sub_batch:
        ch.set_sub_batch(sub_batch)
    return ch


def apply_setter(ch, rdsp, op):
    k = op[0]
    if k == "als":
        if op[1] == "off":
            ch.disableALSfilter()
        else:
            ch.enableALSfilter()
            ch.setALSfilterNotch() if op[1] == "notch" else ch.setALSfilterPeak()
    elif k == "nr":
        ch.set_nr_level(op[1])
    elif k == "spec":
        ch.set_spectral_nr(op[1], op[2])
    elif k == "agc":
        ch.setAGCmode(rdsp.AGC[op[1]])
    elif k == "gdemod":
        ch.group_setDemodMode(op[1], rdsp.DEMOD[op[2]])
    elif k == "gfilt":
        ch.group_reInitializeFilter(op[1], op[2], op[3])
    elif k == "gpbt":
        ch.group_pbt(op[1], op[2], op[3])
    elif k == "goff":
        ch.group_setTuningOffsetHz(op[1], op[2])
    elif k == "gaf":
        ch.group_setAudioFilter(op[1], op[2])
    elif k == "nb":
        ch.enableNoiseBlanker() if op[1] else ch.disableNoiseBlanker()
        ch.setNoiseBlankerThresholdDb(op[2])
    elif k == "swap":
        ch.swapIQ(op[1])
    elif k == "slip":
        ch.setIQslip(op[1])
    elif k == "mute":
        ch.setMute(op[1])
    elif k == "ogain":
        ch.setOutputGain(op[1])
    elif k == "igain":
        ch.setInputGain(op[1])
    elif k == "bal":
        ch.setIQgainBalance(op[1])
    elif k == "afk":
        ch.setAudioFilterKind(op[1])
    elif k == "specform":
        ch.set_spectral_resynthesis(op[1])
    elif k == "emode":
        ch.set_nlms_energy_mode(op[1])
    else:
        raise ValueError(op)


def run_ops(ch, rdsp, torch, ops, dev, out, pos=0):
    """ops on one chain; audio of the calls into `out` at the stream position; returns the new position"""
    for op in ops:
        if op[0] == "proc":
            n = op[1]
            ch.process(dev[:, pos * 128:(pos + n) * 128], out=out[:, pos * 32:(pos + n) * 32])
            pos += n
        else:
            apply_setter(ch, rdsp, op)
    return pos


def run_script(rdsp, torch, ops, iq, n_channels, pipelined, sub_batch, cfg=K3):
    ch = make_chain(n_channels, pipelined, sub_batch, cfg)
    total = sum(op[1] for op in ops if op[0] == "proc")
    out = torch.zeros((n_channels, total * 32, 2), dtype=torch.int16, device="cuda")
    dev = torch.from_numpy(np.ascontiguousarray(iq[:n_channels])).cuda()
    run_ops(ch, rdsp, torch, ops, dev, out)
    ch.flush()
    torch.cuda.synchronize()
    return out.cpu().numpy(), ch.scalars(), ch.lms_coeffs(0), ch.lms_coeffs(1)


@pytest.mark.parametrize("seed", [41, 42, 43, 44, 45, 46])
def test_random_session_resumed_from_a_checkpoint_is_bit_exact(rdsp, seed):
    """A script cut at a random call: the first part on one chain, rdsp_chain_save_state, then a fresh chain
    that is given the same settings (the setters of the first part, no calls), rdsp_chain_load_state and
    the rest of the script -- against the uninterrupted run: audio of the rest and the final state blob,
    bit for bit.  Whatever the record leaves out (a tuning change right before the cut, a blanker level,
    the swap flag the history came in with) shows up here."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from radiodsp_sdr_rx_amd.chain import synth_iq
    rng = np.random.default_rng(seed)
    ops = make_script(rng)
    procs = [i for i, op in enumerate(ops) if op[0] == "proc"]
    if len(procs) < 3:                                         # (a soak seed with two calls: cut between them)
        pytest.skip("script with fewer than three calls") if len(procs) < 2 else None
    cut = procs[int(rng.integers(1, max(2, len(procs) - 1)))] + 1  # right after a call; setters may follow before the next one
    total = sum(op[1] for op in ops if op[0] == "proc")
    nch = 40
    iq = synth_iq(nch, total * 128)
    dev = torch.from_numpy(iq).cuda()
    piped = bool(seed % 2)

    def fresh_out():
        return torch.zeros((nch, total * 32, 2), dtype=torch.int16, device="cuda")

    a, out_a = make_chain(nch, piped, 0), fresh_out()
    run_ops(a, rdsp, torch, ops, dev, out_a)
    a.flush()
    b, out_b = make_chain(nch, piped, 0), fresh_out()
    pos = run_ops(b, rdsp, torch, ops[:cut], dev, out_b)
    blob = b.save_state()
    c, out_c = make_chain(nch, piped, 0), fresh_out()
    for op in ops[:cut]:
        if op[0] != "proc":
            apply_setter(c, rdsp, op)
    c.load_state(blob)
    run_ops(c, rdsp, torch, ops[cut:], dev, out_c, pos)
    c.flush()
    torch.cuda.synchronize()
    assert np.array_equal(out_c.cpu().numpy()[:, pos * 32:], out_a.cpu().numpy()[:, pos * 32:])
    assert np.array_equal(c.save_state(), a.save_state())


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_random_control_session_plain_pipelined_and_repartitioned_agree_bitwise(rdsp, seed):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from radiodsp_sdr_rx_amd.chain import synth_iq
    rng = np.random.default_rng(seed)
    # seeds 9, 10: the four-wave front kernels (FFT_L 2048: a call is a multiple of 32 blocks)
    cfg = dict(K3, fft_l=2048) if seed >= 9 else K3
    ops = make_script(rng, n_ops=40 if seed < 9 else 30, granule=32 if seed >= 9 else 8)
    total = sum(op[1] for op in ops if op[0] == "proc")
    iq = synth_iq(NCH, total * 128)
    # a few rail-to-rail bursts so that an enabled blanker has something to do
    for c in range(0, NCH, 7):
        for p in rng.integers(2000, iq.shape[1] - 4, 12):
            iq[c, p:p + 3] = 30000
    plain = run_script(rdsp, torch, ops, iq, NCH, False, 0, cfg)
    piped = run_script(rdsp, torch, ops, iq, NCH, True, 64, cfg)      # 64 + 64 + 22 channels per stage launch
    names = ("audio", "scalars", "DSP-NR weights", "ALS weights")
    # equal_nan: the reference's NLMS (arm_lms_norm_f32: energy by running difference) can blow up after
    # a loud-to-quiet transition -- the literal CPU restatement does the same -- and a
    # script that tunes a group off its signal produces exactly that; both chains must then blow up alike
    for name, a, b in zip(names, plain, piped):
        assert np.array_equal(a, b, equal_nan=True), f"seed {seed}: {name} differ between the plain and the pipelined chain"
    assert np.isfinite(plain[1]).all()
    # 63 = 21 x 3 keeps the channel -> group map (c mod 3) of the first channels
    part = run_script(rdsp, torch, ops, iq, 63, True, 0, cfg)
    for name, a, b in zip(names, plain, part):
        assert np.array_equal(a[:63], b, equal_nan=True), f"seed {seed}: {name} depend on the channel partition"


# ---- the same idea against the oracle: feed-forward chains, 1e-5 -------------------------------
def _oracle_script(rng, oracle, rdsp, n_groups, granule, n_ops=28):
    """per-group mode changes (tuningMode()'s pair: setAudioFilter + setDemodMode, CTL:330-423), retunes,
    PBT steps and offsets, global swap / blanker / gains / mute, calls of random length.  Every op
    carries what the oracle chain of a channel of that group has to be told."""
    from radiodsp_sdr_rx_amd.chain import pbt_step
    ops = []
    band = [(300.0, 2700.0)] * n_groups
    gains = dict(input_gain=1.0, iq_balance=1.0, output_gain=0.5, mute=False)
    for _ in range(n_ops):
        kind = str(rng.choice(["proc", "proc", "proc", "dcp", "mode", "filt", "pbt", "nco", "swap", "slip", "nb", "gain", "agc", "spec", "specform"]))
        g = int(rng.integers(0, n_groups))
        if kind == "proc":
            ops.append(("proc", granule * int(rng.integers(1, 4))))
        elif kind == "dcp":     # the reference-shaped call, bFilterEnabled per call (CONV:228,300)
            ops.append(("dcp", granule * int(rng.integers(1, 3)), bool(rng.integers(0, 2))))
        elif kind == "mode":
            filt, demod = int(rng.integers(0, 5)), str(rng.choice(["USB", "LSB", "CW_USB", "CW_LSB", "AM"]))
            band[g] = oracle.passband(filt, rdsp.DEMOD[demod])
            ops.append(("mode", g, filt, demod) + band[g])
        elif kind == "filt":
            lo = float(rng.choice([100.0, 300.0, 600.0]))
            band[g] = (lo, lo + float(rng.choice([400.0, 1800.0, 2400.0])))
            ops.append(("filt", g) + band[g])
        elif kind == "pbt":
            if band[g][0] < 0.0:      # the sketch steps its positive dFLoCut / dFHiCut (CTL:569-612)
                continue
            edge, d = int(rng.integers(0, 2)), int(rng.choice([-1, 1]))
            band[g] = pbt_step(band[g][0], band[g][1], edge, d)
            ops.append(("pbt", g, edge, d) + band[g])
        elif kind == "nco":
            ops.append(("nco", g, float(rng.choice([12000.0, 11300.0, 12650.0]))))
        elif kind == "swap":
            ops.append(("swap", bool(rng.integers(0, 2))))
        elif kind == "slip":
            ops.append(("slip", int(rng.choice([-1, 0, 1]))))
        elif kind == "nb":
            ops.append(("nb", bool(rng.integers(0, 2)), float(rng.choice([8.0, 12.0]))))
        elif kind == "agc":
            ops.append(("agc", str(rng.choice(["off", "fast", "medium", "slow"]))))
        elif kind == "spec":
            ops.append(("spec", int(rng.choice([0, 1, 2])), float(rng.choice([1.0, 2.0, 3.0]))))
        elif kind == "specform":
            ops.append(("specform", bool(rng.integers(0, 2))))
        elif kind == "gain":
            which = str(rng.choice(["input_gain", "iq_balance", "output_gain", "mute"]))
            gains[which] = {"input_gain": float(rng.choice([0.5, 1.0, 1.4])), "iq_balance": float(rng.choice([1.0, 1.02, 0.96])),
                            "output_gain": float(rng.choice([0.25, 0.5, 1.0])), "mute": bool(rng.integers(0, 3) == 0)}[which]
            ops.append(("gain", dict(gains)))
    ops.append(("gain", dict(gains, mute=False)))
    ops.append(("proc", 2 * granule))
    return ops


@pytest.mark.parametrize("seed", list(range(11, 27)))
def test_random_retune_session_matches_oracle(rdsp, oracle, seed):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from cases import TOL
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    rng = np.random.default_rng(seed)
    n_groups, nch = 3, 6
    base = dict(fft_l=int(rng.choice([256, 512, 1024, 2048])), agc_mode=str(rng.choice(["off", "medium", "fast"])),
                spectral_nr=int(rng.choice([0, 1])), spectral_level=2.0, output_gain=0.5,
                window=int(rng.choice([1, 2, 4])))
    granule = max(8, base["fft_l"] // 64)   # an overlap-save hop of FFT_L / 2 samples at 24 kHz and the front kernel's 1024-sample chunk, in 128-sample blocks
    ops = _oracle_script(rng, oracle, rdsp, n_groups, granule)
    total = sum(op[1] for op in ops if op[0] in ("proc", "dcp"))
    iq = synth_iq(nch, total * 128)
    for c in range(nch):
        for p in rng.integers(3000, iq.shape[1] - 4, 10):
            iq[c, p:p + 3] = rng.choice([-30000, 30000])
    group_of = (np.arange(nch) % n_groups).astype(np.uint16)
    ch = Chain(nch, max_blocks_per_call=3 * granule, **base)
    ch.set_groups(group_of)
    ch.set_pipelined(bool(seed % 2))
    if seed % 3 == 0:
        ch.set_fir_variant(0)        # the direct-form decimator keeps its own copy of the history path
    ocs = [oracle.OracleChain(**base) for _ in range(nch)]
    dev = torch.from_numpy(iq).cuda()
    got, ref, pos = [], [[] for _ in range(nch)], 0
    spectral_now = base["spectral_nr"] != 0
    for op in ops:
        k = op[0]
        if k == "spec":
            spectral_now = op[1] != 0
        mine = [c for c in range(nch) if k in ("mode", "filt", "pbt", "nco") and group_of[c] == op[1]]
        if k == "proc":
            n = op[1]
            got.append(ch.process(dev[:, pos * 128:(pos + n) * 128], want_f32=True)[1])
            for c in range(nch):
                ref[c].append(ocs[c].process(iq[c, pos * 128:(pos + n) * 128])[1])
            pos += n
        elif k == "dcp":        # doConvolutionalProcessing(nr_level, bFilterEnabled, lo, hi): int16 audio only
            n, f = op[1], op[2]
            o16 = ch.doConvolutionalProcessing(0, f, 0.0, 0.0, dev[:, pos * 128:(pos + n) * 128])
            ch.flush()
            torch.cuda.synchronize()
            o16 = o16.cpu().numpy()
            for c in range(nch):
                ocs[c].set_filter_on(f)
                r16, r32 = ocs[c].process(iq[c, pos * 128:(pos + n) * 128])
                d = np.abs(o16[c].astype(np.int32) - r16.astype(np.int32)).max(axis=1)
                lim = max(1, int(np.ceil(2 * TOL * 32768 * np.abs(r32).max())))
                hops = d[:len(d) // (base["fft_l"] // 2) * (base["fft_l"] // 2)].reshape(-1, base["fft_l"] // 2).max(axis=1)
                if spectral_now:   # the threshold discontinuity of the spectral stage, see the comment at the end
                    assert (hops > lim).sum() <= 2 and hops.max() <= 164, (seed, op, c, hops)   # 5e-3 of full scale
                else:
                    assert hops.max() <= lim, (seed, op, c, hops.max())
                ref[c].append(r32)
            got.append(torch.zeros((nch, n * 128 // 4, 2), dtype=torch.float32, device="cuda") + float("nan"))
            pos += n
        elif k == "mode":
            ch.group_setAudioFilter(op[1], op[2])
            ch.group_setDemodMode(op[1], rdsp.DEMOD[op[3]])
            for c in mine:
                ocs[c].set_demod(rdsp.DEMOD[op[3]])
                ocs[c].reinit_filter(op[4], op[5])
        elif k == "filt":
            ch.group_reInitializeFilter(op[1], op[2], op[3])
            for c in mine:
                ocs[c].reinit_filter(op[2], op[3])
        elif k == "pbt":
            ch.group_pbt(op[1], op[2], op[3])
            for c in mine:
                ocs[c].reinit_filter(op[4], op[5])
        elif k == "nco":
            ch.group_setTuningOffsetHz(op[1], op[2])
            for c in mine:
                ocs[c].set_nco_hz(op[2])
        elif k == "swap":
            ch.swapIQ(op[1])
            for oc in ocs:
                oc.set_swap_iq(op[1])
        elif k == "slip":
            ch.setIQslip(op[1])
            for oc in ocs:
                oc.set_iq_slip(op[1])
        elif k == "nb":
            ch.enableNoiseBlanker() if op[1] else ch.disableNoiseBlanker()
            ch.setNoiseBlankerThresholdDb(op[2])
            for oc in ocs:
                oc.set_noise_blanker(op[1], op[2])
        elif k == "agc":
            ch.setAGCmode(rdsp.AGC[op[1]])
            for oc in ocs:
                oc.set_agc_mode(rdsp.AGC[op[1]])
        elif k == "spec":
            ch.set_spectral_nr(op[1], op[2])
            for oc in ocs:
                oc.set_spectral_nr(op[1], op[2])
        elif k == "specform":   # SPEC:229-232 as written on both sides, or the equivalent form on both
            ch.set_spectral_resynthesis(op[1])
            for oc in ocs:
                oc.set_literal_resynthesis(op[1])
        elif k == "gain":
            gn = op[1]
            ch.setInputGain(gn["input_gain"]); ch.setIQgainBalance(gn["iq_balance"])
            ch.setOutputGain(gn["output_gain"]); ch.setMute(gn["mute"])
            for oc in ocs:
                oc.set_gains(gn["input_gain"], gn["iq_balance"], gn["output_gain"], gn["mute"])
    ch.flush()
    torch.cuda.synchronize()
    got = np.concatenate([o.cpu().numpy() for o in got], 1)
    hop = base["fft_l"] // 2
    for c in range(nch):
        r = np.concatenate(ref[c])
        e = np.abs(got[c] - r).max(axis=1) / np.abs(r).max()
        e = np.where(np.isnan(e), 0.0, e)      # the reference-shaped calls were checked on their int16 audio
        per_hop = e[:len(e) // hop * hop].reshape(-1, hop).max(axis=1)
        bad = per_hop > TOL
        if base["spectral_nr"] or any(op[0] == "spec" and op[1] for op in ops):
            # the spectral stage is discontinuous at its threshold (SPEC:213-217: 0.2 mag below NFloor,
            # mag - NFloor above): a bin within rounding of NFloor lands on either side, which moves one
            # bin of one frame by 0.2 NFloor, i.e. one hop of output by ~1e-4 of full scale.  2048 bins x
            # every frame x every channel make that a one-in-ten event per session; it is confined to
            # that hop (every output sample comes from exactly one frame).  Everything else is at 1e-5.
            # (the largest such hop in a soak of 560 sessions: 1.9e-3; a semantic error is of order one)
            assert bad.sum() <= 2 and per_hop.max() <= 5e-3, f"seed {seed} ({base}), channel {c}: {per_hop[bad]}\n{ops}"
        else:
            assert not bad.any(), f"seed {seed} ({base}), channel {c}: {per_hop.max():.2e}\n{ops}"


@pytest.mark.parametrize("seed", [31, 32, 33, 34, 35, 36])
def test_random_nr_and_notch_switching_is_truth_anchored(rdsp, oracle, seed):
    """The recursive stages switched at random between calls: ALS off / notch / peak, DSP-NR level
    0 / 20 / 30 (a level change re-initialises the instance, CONV:327-331), AGC modes, the spectral stage.
    Three runs of the same script: GPU, float32 oracle, float64 model (tests/np_model.py takes the same
    switches); the criterion is the truth-anchored one of the NLMS chains -- the GPU no further from
    the float64 result than max(1e-5, 1.5 x the oracle's own distance).  What keeps its state, what is
    re-initialised and which call a change takes effect in are all inside that bound: a wrong answer to
    any of them is an error of order one."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import np_model
    from oracle_lib import AGC, ALS
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from test_gpu_parity import assert_truth_anchored
    rng = np.random.default_rng(seed)
    nch = 5
    base = dict(K3, fft_l=int(rng.choice([256, 512])))
    ops = []
    for _ in range(22):
        kind = str(rng.choice(["proc", "proc", "als", "nr", "agc", "spec"]))
        if kind == "proc":
            ops.append(("proc", 8 * int(rng.integers(1, 4))))
        elif kind == "als":
            ops.append(("als", str(rng.choice(["off", "notch", "peak"]))))
        elif kind == "nr":
            ops.append(("nr", int(rng.choice([0, 20, 30]))))
        elif kind == "agc":
            ops.append(("agc", str(rng.choice(["off", "fast", "medium", "slow"]))))
        else:
            ops.append(("spec", int(rng.choice([0, 1])), float(rng.choice([1.0, 2.0]))))
    ops.append(("proc", 16))
    total = sum(op[1] for op in ops if op[0] == "proc")
    iq = synth_iq(nch, total * 128)
    ch = Chain(nch, max_blocks_per_call=24, **base)
    ch.set_pipelined(bool(seed % 2))
    ocs = [oracle.OracleChain(**base) for _ in range(nch)]
    mods = [np_model.Model(**base) for _ in range(nch)]
    dev = torch.from_numpy(iq).cuda()
    got, ref, f64, pos = [], [[] for _ in range(nch)], [[] for _ in range(nch)], 0
    for op in ops:
        k = op[0]
        if k == "proc":
            n = op[1]
            got.append(ch.process(dev[:, pos * 128:(pos + n) * 128], want_f32=True)[1])
            for c in range(nch):
                ref[c].append(ocs[c].process(iq[c, pos * 128:(pos + n) * 128])[1])
                f64[c].append(mods[c].process(iq[c, pos * 128:(pos + n) * 128]))
            pos += n
        elif k == "als":
            if op[1] == "off":
                ch.disableALSfilter()
            else:
                ch.enableALSfilter()
                ch.setALSfilterNotch() if op[1] == "notch" else ch.setALSfilterPeak()
            for oc, m in zip(ocs, mods):
                oc.set_als_mode(rdsp.ALS[op[1]])
                m.c["als_mode"] = ALS[op[1]]
        elif k == "nr":
            ch.set_nr_level(op[1])
            for oc, m in zip(ocs, mods):
                oc.set_nr_level(op[1])
                m.c["lms_nr"] = op[1]
        elif k == "agc":
            ch.setAGCmode(rdsp.AGC[op[1]])
            for oc, m in zip(ocs, mods):
                oc.set_agc_mode(rdsp.AGC[op[1]])
                m.c["agc_mode"] = AGC[op[1]]
        elif k == "spec":
            ch.set_spectral_nr(op[1], op[2])
            for oc, m in zip(ocs, mods):
                oc.set_spectral_nr(op[1], op[2])
                m.c["spectral_nr"], m.c["spectral_level"] = op[1], op[2]
    ch.flush()
    torch.cuda.synchronize()
    got = np.concatenate([o.cpu().numpy() for o in got], 1)
    r32 = np.stack([np.concatenate(r) for r in ref])
    t64 = np.stack([np.concatenate(r) for r in f64])
    assert np.isfinite(r32).all() and np.isfinite(got).all()
    assert_truth_anchored(got, r32, t64, f"seed {seed}")
