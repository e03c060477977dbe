"""the per-channel DSP state as data.  The reference keeps it in globals
(RDSP_convolutional.h:50-57,77-80; RDSP_noise_reduction.h:26-32; RDSP_convolutional_spec.h:109) and has no
persistence; rdsp_chain_save_state / rdsp_chain_load_state write a range of channels out and read it back
into another chain: resume after a restart, channels moved between shards or GPUs.  Everything here is
bit-exact: a stream continued from a blob is the uninterrupted stream."""
import numpy as np
import pytest

from cases import K1, K3

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("front_form")]   # every test under both front kernels (conftest.py)


def _parts(iq, per):
    import torch
    return [torch.from_numpy(np.ascontiguousarray(iq[:, k * per * 128:(k + 1) * per * 128])).cuda()
            for k in range(iq.shape[1] // (per * 128))]


def _setup(ch, kind):
    if kind == "sam_iir":           # the optional arrays: SAM PLL state, IIR cascade state
        ch.setAudioFilterKind(1)
        ch.setAudioFilter(4)
        ch.setDemodMode(6)          # SAM
    elif kind == "k3_nr":           # both NLMS instances, blanker level, swap / gains carried as history parameters
        ch.set_nr_level(20)
        ch.enableNoiseBlanker()
        ch.swapIQ(True)
        ch.setInputGain(0.8)
    elif kind == "k1_slip":         # the I2S slip correction: its carry word (the last raw sample) travels too
        ch.setIQslip(1)


@pytest.mark.parametrize("kind,cfg", [("k3", K3), ("k3_nr", K3), ("sam_iir", dict(fft_l=512, agc_mode="slow", output_gain=0.5)),
                                      ("k1", K1), ("k1_slip", K1), ("cw_2048", dict(fft_l=2048, demod="CW_USB", flo_hz=450.0, fhi_hz=950.0,
                                                                    nco_hz=11300.0, agc_mode="fast"))])
def test_resume_and_channel_move_are_bit_exact(rdsp, kind, cfg):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    nch, per = 11, (32 if cfg.get("fft_l", 256) >= 2048 else 16)
    iq = synth_iq(nch, per * 4 * 128)
    parts = _parts(iq, per)

    def new(n):
        ch = Chain(n, max_blocks_per_call=per, **cfg)
        _setup(ch, kind)
        ch.set_pipelined(True)
        return ch

    def run(ch, ps):                               # pipelined: the audio is complete at flush()
        outs = [ch.process(p) for p in ps]
        ch.flush()
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in outs]

    a = new(nch)                                   # the uninterrupted stream
    ref = run(a, parts)
    b = new(nch)                                   # stops after two calls
    for p in parts[:2]:
        b.process(p)
    blob = b.save_state()
    assert blob.size == rdsp.load().rdsp_chain_state_bytes(b.h, nch)
    part = b.save_state(3, 4)                      # channels 3..6 only
    c = new(nch)                                   # restart: a fresh chain, same settings
    c.load_state(blob)
    for g, r in zip(run(c, parts[2:]), ref[2:]):
        assert np.array_equal(g, r)
    assert np.array_equal(c.scalars(), a.scalars()) and np.array_equal(c.lms_coeffs(1), a.lms_coeffs(1))
    assert np.array_equal(c.save_state(), a.save_state())      # the whole record, not only what the getters show
    d = new(4)                                     # those four channels on another chain (another shard / GPU)
    d.load_state(part)
    for g, r in zip(run(d, [p[3:7].contiguous() for p in parts[2:]]), ref[2:]):
        assert np.array_equal(g, r[3:7])
    # ... and back into a running chain at the same stream position, other channels untouched
    e = new(nch)
    for p in parts[:2]:
        e.process(p)
    e.load_state(part, first_channel=3)
    assert np.array_equal(e.save_state(), blob)


def test_state_argument_errors_are_loud(rdsp):
    import torch
    assert torch.cuda.is_available()
    from radiodsp_sdr_rx_amd import RdspError
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    a = Chain(4, max_blocks_per_call=8, **K1)
    blob = a.save_state()
    with pytest.raises(RdspError):
        Chain(4, max_blocks_per_call=8, **dict(K1, fft_l=512)).load_state(blob)       # another FFT_L
    with pytest.raises(RdspError):
        Chain(3, max_blocks_per_call=8, **K1).load_state(blob)                         # does not fit
    with pytest.raises(RdspError):
        a.load_state(blob[:-8])                                                        # truncated
    bad = blob.copy(); bad[0] ^= 0xFF
    with pytest.raises(RdspError):
        a.load_state(bad)                                                              # not a state blob
    with pytest.raises(RdspError):
        a.save_state(2, 3)                                                             # range past the end
    b = Chain(4, max_blocks_per_call=8, **K1)
    b.process(torch.from_numpy(synth_iq(4, 8 * 128)).cuda())
    with pytest.raises(RdspError):
        b.load_state(blob)                                                             # a running chain at another stream position
    # nothing is restored in part: optional state in the blob needs its place in the chain
    s = Chain(4, max_blocks_per_call=8, **K1)
    s.setIQslip(-1)
    s.process(torch.from_numpy(synth_iq(4, 8 * 128)).cuda())
    with pytest.raises(RdspError):
        Chain(4, max_blocks_per_call=8, **K1).load_state(s.save_state())               # slip carry without setIQslip
    i = Chain(4, max_blocks_per_call=8, **K1)
    i.setAudioFilterKind(1)
    i.process(torch.from_numpy(synth_iq(4, 8 * 128)).cuda())
    with pytest.raises(RdspError):
        Chain(4, max_blocks_per_call=8, **K1).load_state(i.save_state())               # IIR cascade state without the IIR bank
    ok = Chain(4, max_blocks_per_call=8, **K1)
    ok.setAudioFilterKind(1)
    ok.load_state(i.save_state())


def test_a_stream_continues_in_another_decimator_form(rdsp, oracle):
    """All forms of stage A3 keep the same state (the last 256 raw samples, the previous hop of the decimated
    stream), so a stream saved under one may be continued under another -- a recording processed in the throughput
    form (rdsp_chain_set_fir_variant 2) up to a checkpoint and resumed by a host that wants split-invariant bits, or
    the other way round.  The two forms round differently, so this is a tolerance statement: every hand-over follows
    the oracle's uninterrupted stream at 1e-5."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from cases import TOL
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    cfg = dict(fft_l=512, demod="USB", spectral_nr=1, spectral_level=2.0, agc_mode="medium")
    nch, per = 3, 16
    iq = synth_iq(nch, per * 4 * 128)
    parts = _parts(iq, per)
    ref = np.stack([oracle.OracleChain(**cfg).process(iq[c])[1] for c in range(nch)])
    # -1: the default (no tail stage in this chain: the row form, 5 by name); 4: one granule per wave-wide frame
    for first, second in ((2, -1), (-1, 2), (0, -1), (2, 0), (4, 5), (5, 4), (-1, 0)):
        a = Chain(nch, max_blocks_per_call=per, fir_variant=first, **cfg)
        got = [a.process(p, want_f32=True)[1].cpu().numpy() for p in parts[:2]]
        b = Chain(nch, max_blocks_per_call=per, fir_variant=second, **cfg)
        b.load_state(a.save_state())
        got += [b.process(p, want_f32=True)[1].cpu().numpy() for p in parts[2:]]
        got = np.concatenate(got, 1)
        err = max(np.abs(got[c] - ref[c]).max() / np.abs(ref[c]).max() for c in range(nch))
        assert err <= TOL, (first, second, err)
