"""SURVEY 8f row F3 (first part): engine features that live in the AudioSDR library and
are therefore build-defined (stated in DESIGN.md 6e and in the oracle): the
pre-processor's IQ swap (INO:118) and the noise blanker (BK_INO:1259-1260, INO:131).
GPU through the C-ABI against the oracle; tolerance TOL = 1e-5 normwise per channel."""
import numpy as np
import pytest

from cases import K1, K3, TOL
from parity_util import assert_truth_anchored, model_run, normwise

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("front_form")]   # every test under both front kernels (conftest.py)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _impulses(iq, seed=4, per_channel=24, burst=3):
    """ignition-noise style bursts: a few full-scale samples, far above any threshold"""
    rng = np.random.default_rng(seed)
    out = iq.copy()
    for c in range(iq.shape[0]):
        for pos in rng.integers(3000, iq.shape[1] - burst, per_channel):
            out[c, pos:pos + burst] = rng.choice([-30000, 30000], size=(burst, 2))
    return out


def test_swap_iq_matches_oracle_and_is_a_pure_relabelling(rdsp, oracle, torch_cuda):
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    iq = synth_iq(3, 32 * 128)
    cfg = dict(K1, fft_l=512, iq_balance=1.02)
    ch = Chain(3, max_blocks_per_call=32, **cfg)
    ch.startAutoI2SerrorDetection()                  # accepted, nothing to detect without an I2S bus
    ch.swapIQ(True)
    got16, got = ch.process(torch.from_numpy(iq).cuda(), want_f32=True)
    got = got.cpu().numpy()
    for c in range(3):
        oc = oracle.OracleChain(**cfg)
        oc.set_swap_iq(True)
        ref = oc.process(iq[c])[1]
        assert np.abs(got[c] - ref).max() / np.abs(ref).max() <= TOL
    plain = Chain(3, max_blocks_per_call=32, **cfg).process(torch.from_numpy(np.ascontiguousarray(iq[..., ::-1])).cuda())
    assert np.array_equal(got16.cpu().numpy(), plain.cpu().numpy())


def test_iq_slip_correction_matches_oracle_and_undoes_a_slipped_recording(rdsp, oracle, torch_cuda):
    """INO:117 guards against an I2S fault that leaves one rail a sample late; a recording made through such
    a front end carries it.  rdsp_pre_setIQslip pairs I[n-1] with Q[n] (+1) or I[n] with Q[n-1] (-1) on the
    raw words, before swapIQ and the gains, on samples as they arrive (build-defined; the oracle states the
    same).  Switched between calls (off -> +1 -> -1 -> off, with swapIQ on) it follows the oracle at TOL;
    a recording whose Q rail is one sample late, corrected with +1, is bit for bit the clean stream one
    sample later; pipelined and split calls are bit-identical; the host estimate finds the slip."""
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, estimate_iq_slip, synth_iq
    nch, per, calls = 3, 16, 4
    iq = synth_iq(nch, per * calls * 128)
    cfg = dict(K1, fft_l=512, iq_balance=1.02)
    script = [0, 1, -1, 0]                              # the slip each call runs with
    ch = Chain(nch, max_blocks_per_call=per, **cfg)
    ch.swapIQ(True)
    ocs = [oracle.OracleChain(**cfg) for _ in range(nch)]
    for oc in ocs:
        oc.set_swap_iq(True)
    got, ref = [], [[] for _ in range(nch)]
    for k, sl in enumerate(script):
        part = iq[:, k * per * 128:(k + 1) * per * 128]
        ch.setIQslip(sl)
        got.append(ch.process(torch.from_numpy(np.ascontiguousarray(part)).cuda(), want_f32=True)[1].cpu().numpy())
        for c, oc in enumerate(ocs):
            oc.set_iq_slip(sl)
            ref[c].append(oc.process(part[c])[1])
    got = np.concatenate(got, 1)
    for c in range(nch):
        r = np.concatenate(ref[c])
        assert np.abs(got[c] - r).max() / np.abs(r).max() <= TOL
    # the same script on a chain without the decimator (the literal CONV stage at its native rate): there is no
    # FIR history there, and the word in front of a call's first sample must still be the previous call's last one
    from cases import CONV_LITERAL
    lit = Chain(nch, max_blocks_per_call=per, **CONV_LITERAL)
    olit = [oracle.OracleChain(**CONV_LITERAL) for _ in range(nch)]
    got, ref = [], [[] for _ in range(nch)]
    for k, sl in enumerate(script):
        part = iq[:, k * per * 128:(k + 1) * per * 128]
        lit.setIQslip(sl)
        got.append(lit.process(torch.from_numpy(np.ascontiguousarray(part)).cuda(), want_f32=True)[1].cpu().numpy())
        for c, oc in enumerate(olit):
            oc.set_iq_slip(sl)
            ref[c].append(oc.process(part[c])[1])
    got = np.concatenate(got, 1)
    for c in range(nch):
        r = np.concatenate(ref[c])
        assert np.abs(got[c] - r).max() / np.abs(r).max() <= TOL, ("decim 1", c)
    # a recording with the fault: Q one sample late.  Corrected it is the clean stream delayed by one sample.
    late = iq.copy()
    late[:, 1:, 1] = iq[:, :-1, 1]
    late[:, 0, 1] = 0
    delayed = np.zeros_like(iq)
    delayed[:, 1:] = iq[:, :-1]
    assert estimate_iq_slip(late[0])[0] == 1 and estimate_iq_slip(iq[0])[0] == 0
    a = Chain(nch, max_blocks_per_call=per * calls, **K3)
    a.setIQslip(1)
    b = Chain(nch, max_blocks_per_call=per * calls, **K3)
    oa = a.process(torch.from_numpy(late).cuda()).cpu().numpy()
    ob = b.process(torch.from_numpy(delayed).cuda()).cpu().numpy()
    assert np.array_equal(oa, ob)
    # split + pipelined calls carry the last raw word from call to call: the same bits
    p = Chain(nch, max_blocks_per_call=per, **K3)
    p.setIQslip(1)
    p.set_pipelined(True)
    outs = [p.process(torch.from_numpy(np.ascontiguousarray(late[:, k * per * 128:(k + 1) * per * 128])).cuda()) for k in range(calls)]
    p.flush()
    torch.cuda.synchronize()
    q = Chain(nch, max_blocks_per_call=per, **K3)
    q.set_fir_variant(0)
    q.setIQslip(1)
    outs_q = [q.process(torch.from_numpy(np.ascontiguousarray(late[:, k * per * 128:(k + 1) * per * 128])).cuda()) for k in range(calls)]
    one = Chain(nch, max_blocks_per_call=per * calls, **K3)
    one.set_fir_variant(0)
    one.setIQslip(1)
    assert np.array_equal(np.concatenate([o.cpu().numpy() for o in outs_q], 1), one.process(torch.from_numpy(late).cuda()).cpu().numpy())
    w = Chain(nch, max_blocks_per_call=per, **K3)      # un-pipelined, same call split as p
    w.setIQslip(1)
    outs_w = [w.process(torch.from_numpy(np.ascontiguousarray(late[:, k * per * 128:(k + 1) * per * 128])).cuda()) for k in range(calls)]
    assert all(np.array_equal(x.cpu().numpy(), y.cpu().numpy()) for x, y in zip(outs, outs_w))


@pytest.mark.parametrize("name,cfg", [("k1", K1), ("k3_front", dict(K3, als_mode="off")), ("literal", None),
                                      # four waves per channel: the blanker's pre-pass goes round the waves in frame order
                                      ("usb_2048", dict(fft_l=2048, demod="USB", agc_mode="medium", output_gain=0.5)),
                                      ("k4_4096", "K4")])
def test_noise_blanker_matches_oracle(rdsp, oracle, torch_cuda, front_form, name, cfg):
    torch = torch_cuda
    from cases import CONV_LITERAL, K4
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    cfg = CONV_LITERAL if cfg is None else (K4 if cfg == "K4" else cfg)
    nch, nblk = 4, (128 if cfg.get("fft_l", 256) >= 2048 else 64)
    clean = synth_iq(nch, nblk * 128)
    iq = _impulses(clean)
    ch = Chain(nch, max_blocks_per_call=nblk, **cfg)
    ch.enableNoiseBlanker()
    ch.setNoiseBlankerThresholdDb(8.0)
    dev = torch.from_numpy(iq).cuda()
    got = ch.process(dev, want_f32=True)[1].cpu().numpy()
    if cfg.get("decim", 4) == 4:      # the frequency-domain kernel whatever FFT_L: no fall-back to the direct form
        assert ch.front_kernel_name() == ("rdsp_front_kernel" if front_form == "direct" else "rdsp_front_fd_kernel")
    levels = ch.scalars()[:, 3]
    off = Chain(nch, max_blocks_per_call=nblk, **cfg).process(dev, want_f32=True)[1].cpu().numpy()
    ref_clean = Chain(nch, max_blocks_per_call=nblk, **cfg).process(torch.from_numpy(clean).cuda(), want_f32=True)[1].cpu().numpy()
    for c in range(nch):
        oc = oracle.OracleChain(**cfg)
        oc.set_noise_blanker(True, 8.0)
        ref = oc.process(iq[c])[1]
        err = np.abs(got[c] - ref).max() / np.abs(ref).max()
        assert err <= TOL, f"channel {c}: {err:.2e}"
        assert abs(levels[c] - oc.nb_level()) <= 1e-5 * oc.nb_level()
        # it does its job: with the blanker the audio is closer to the impulse-free audio
        # (the AGC-free configs; with AGC the gains differ and the comparison is not meaningful)
        if "agc_mode" not in cfg:
            e_on = np.abs(got[c] - ref_clean[c])[200:].max()
            e_off = np.abs(off[c] - ref_clean[c])[200:].max()
            assert e_on < 0.75 * e_off, (e_on, e_off)   # a zeroed sample leaves a hole of its own


def test_noise_blanker_split_calls_and_pipelined_are_bitwise_identical(rdsp, torch_cuda):
    """a blanked sample stays blanked when it becomes FIR history of the next call"""
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    nch, per, calls = 3, 8, 6
    iq = _impulses(synth_iq(nch, per * calls * 128), per_channel=80)
    # bursts right at call boundaries
    for k in range(1, calls):
        iq[:, k * per * 128 - 2:k * per * 128 + 1] = 30000

    def run(n_calls, pipelined, lean, fir=0):
        ch = Chain(nch, max_blocks_per_call=per * calls // n_calls, **K3)
        ch.enableNoiseBlanker()
        ch.setNoiseBlankerThresholdDb(8.0)
        ch.set_pipelined(pipelined)
        ch.set_fir_variant(fir)     # 0: direct-form decimator, the one that does not depend on the call split
        if lean:
            ch.set_front_variant(0)
        step = iq.shape[1] // n_calls
        outs = [ch.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * step:(k + 1) * step])).cuda())
                for k in range(n_calls)]
        ch.flush()
        torch.cuda.synchronize()
        return np.concatenate([o.cpu().numpy() for o in outs], 1)

    one = run(1, False, True)
    assert np.array_equal(one, run(calls, False, True))
    assert np.array_equal(one, run(calls, True, False))
    # the frequency-domain decimator (opt-in) with the blanker compiled in: the same split,
    # pipelined or not, gives the same bits; another split frames and rounds differently
    fd = run(calls, False, False, fir=2)
    assert np.array_equal(fd, run(calls, True, False, fir=2))
    assert np.abs(fd.astype(np.int32) - one.astype(np.int32)).max() <= 64 and (fd != one).mean() < 0.3


def test_noise_blanker_four_wave_kernels_split_and_pipelined(rdsp, oracle, torch_cuda, front_form):
    """FFT_L 2048 (four waves per channel): bursts at call boundaries and inside the column a frame
    shares with the next one; the same split pipelined or not gives the same bits, a blanked sample
    stays blanked as FIR history, and every call follows the oracle."""
    torch = torch_cuda
    from cases import TOL
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    cfg = dict(fft_l=2048, demod="LSB", flo_hz=-2700.0, fhi_hz=-300.0, agc_mode="fast", output_gain=0.5)
    nch, per, calls = 3, 32, 4
    iq = _impulses(synth_iq(nch, per * calls * 128), per_channel=60)
    for k in range(1, calls):
        iq[:, k * per * 128 - 2:k * per * 128 + 1] = 30000
    for f in range(1, 6):            # the shared column of consecutive decimator frames (1792-sample hop)
        iq[:, f * 1792 - 100:f * 1792 - 97] = -30000

    def run(pipelined):
        ch = Chain(nch, max_blocks_per_call=per, **cfg)
        ch.enableNoiseBlanker()
        ch.setNoiseBlankerThresholdDb(8.0)
        ch.set_pipelined(pipelined)
        outs = [ch.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * per * 128:(k + 1) * per * 128])).cuda(),
                           want_f32=True)[1] for k in range(calls)]
        ch.flush()
        torch.cuda.synchronize()
        assert ch.front_kernel_name() == ("rdsp_front_kernel" if front_form == "direct" else "rdsp_front_fd_kernel")
        return np.concatenate([o.cpu().numpy() for o in outs], 1), ch.scalars()

    a, sa = run(False)
    b, sb = run(True)
    assert np.array_equal(a, b) and np.array_equal(sa, sb)
    for c in range(nch):
        oc = oracle.OracleChain(**cfg)
        oc.set_noise_blanker(True, 8.0)
        ref = np.concatenate([oc.process(iq[c, k * per * 128:(k + 1) * per * 128])[1] for k in range(calls)])
        err = np.abs(a[c] - ref).max() / np.abs(ref).max()
        assert err <= TOL, f"channel {c}: {err:.2e}"
        assert abs(sa[c, 3] - oc.nb_level()) <= 1e-5 * oc.nb_level()


# ---- SAM: PLL synchronous detector (CTL:384-391; build-defined arithmetic) ---------------
def _am_signal(nch, n, fs=96000.0, seed=9, off0=60.0, doff=20.0):
    """AM carriers off0 + c*doff Hz off the 12 kHz tuning offset, 700/1100 Hz modulation, noise"""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / fs
    out = np.empty((nch, n, 2), np.int16)
    for c in range(nch):
        fc = 12000.0 + off0 + doff * c
        m = 0.5 * np.sin(2 * np.pi * 700.0 * t + c) + 0.3 * np.sin(2 * np.pi * 1100.0 * t)
        x = 0.35 * (1.0 + m) * np.exp(2j * np.pi * fc * t + 1j * rng.uniform(0, 6.28))
        x = x + 0.01 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        out[c, :, 0] = np.clip(np.round(x.real * 32767), -32768, 32767)
        out[c, :, 1] = np.clip(np.round(x.imag * 32767), -32768, 32767)
    return out


@pytest.mark.parametrize("pipelined", [False, True])
def test_sam_demodulator_matches_oracle_and_recovers_the_modulation(rdsp, oracle, torch_cuda, pipelined):
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain
    nch, per, calls = 5, 32, 4
    iq = _am_signal(nch, per * calls * 128)
    cfg = dict(fft_l=512, demod="SAM", flo_hz=-3900.0, fhi_hz=3900.0, agc_mode="slow", output_gain=0.5)
    ch = Chain(nch, max_blocks_per_call=per, **cfg)
    ch.set_pipelined(pipelined)
    outs = [ch.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * per * 128:(k + 1) * per * 128])).cuda(),
                       want_f32=True)[1] for k in range(calls)]
    ch.flush()
    torch.cuda.synchronize()
    got = np.concatenate([o.cpu().numpy() for o in outs], 1)
    # a feedback loop fed by the filter's start-up ramp: truth-anchored like the other recursive stages
    # (parity_util.assert_truth_anchored; the float64 PLL is tests/np_model.py)
    ref = np.stack([oracle.OracleChain(**cfg).process(iq[c])[1] for c in range(nch)])
    assert_truth_anchored(got, ref, model_run(iq, cfg), "SAM")
    for c in range(nch):
        a = got[c, 2048:, 0]
        spec = np.abs(np.fft.rfft(a * np.hanning(len(a))))
        f = np.fft.rfftfreq(len(a), 1 / 24000.0)
        assert abs(f[spec.argmax()] - 700.0) < 12.0        # the 700 Hz tone is what comes out


def test_sam_group_beside_other_groups_and_mode_table_entry(rdsp, oracle, torch_cuda):
    torch = torch_cuda
    from radiodsp_sdr_rx_amd.chain import Chain
    nch, nblk = 4, 64
    iq = _am_signal(nch, nblk * 128)
    base = dict(fft_l=512)
    ch = Chain(nch, max_blocks_per_call=nblk, **base)
    ch.set_groups(np.array([0, 1, 0, 1], np.uint16))
    assert ch.group_tuningMode(1, 5, 7.2e6) == 6890         # "SAM": audioAM + SAMmode, CTL:385-392; TuningOffset = the IF centre
    ch.group_tuningMode(0, 4, 7.2e6)                        # "AM"
    got = ch.process(torch.from_numpy(iq).cuda(), want_f32=True)[1].cpu().numpy()
    ok, filt, demod = oracle.tuning_mode(5, 7.2e6)
    lo, hi = oracle.passband(filt, demod)
    assert ok and demod == rdsp.DEMOD["SAM"] and (lo, hi) == (-3900.0, 3900.0)
    for d, sel in (("AM", [0, 2]), ("SAM", [1, 3])):
        cfg = dict(base, demod=d, flo_hz=lo, fhi_hz=hi)
        ref = np.stack([oracle.OracleChain(**cfg).process(iq[c])[1] for c in sel])
        if d == "AM":
            assert normwise(got[sel], ref) <= TOL
        else:
            assert_truth_anchored(got[sel], ref, model_run(iq[sel], cfg), "SAM group")
