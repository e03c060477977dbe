"""The reference's AudioSDR engine (`AudioSDR SDR;`, INO:54) pinned on its own compiled code.

tests/golden/engine_kat.npz was produced in the build container by running AudioSDR::update() of the reference's firmware
image (pre_compiled/RadioDSP_SDR_RX.ino.hex, ITCM 0xe730) under tests/golden/thumb_emu.py, through the sketch's settings
and menus, on seeded int16 IQ (tests/golden/make_engine_kat.py; `--check` reproduces it).  Inputs and outputs only.

`-m "not gpu"`: oracle/rdsp_engine_oracle.c, the restatement written from the image's code, against every case -- the
int16 audio and the float buffers after each stage, BIT FOR BIT -- and the tables it generates against the object's.
`-m gpu`: rdsp_engine_t (csrc/rdsp_engine.hip) through the C-ABI against the same int16 audio, bit for bit as well: the
kernels evaluate the same operations in the same order (the north-star's tolerance for float work is 1e-5; nothing of it
is used here), on one channel per case, on all cases as channels of the same engine where the settings allow, and with
the blocks cut into calls of different sizes."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(HERE, "golden", "engine_kat.npz"))


def case_names(k):
    return [str(n) for n in k["case_names"]]


def calls_of(k, name):
    return json.loads(str(k[name + "_calls"]))


# ---- CPU: the restatement against the image -----------------------------------------------------------------------------
def test_oracle_tables_are_the_objects(kat, oracle):
    """the sine table (eight-place decimals of sin(2 pi k / 256)), the AGC's gain curve (built with the restated newlib
    expf), the PLL's loop gains and setDemodMode's answers, against the object after the image's constructor ran"""
    e = oracle.OracleEngine(sketch_setup=False)
    lib = e.lib
    assert np.array_equal(np.ctypeslib.as_array(lib.orc_engine_sine(e.e), (257,)), kat["sine257"])
    assert np.array_equal(np.ctypeslib.as_array(lib.orc_engine_agc_curve(e.e), (130,))[:129], kat["agc_curve"][:129])
    got = np.array([lib.orc_newlib_expf(float(x)) for x in kat["expf_x"]], np.float32)
    assert np.array_equal(got.view(np.uint32), kat["expf_y"].view(np.uint32))
    assert [e.call("setDemodMode", m) for m in range(7)] == list(kat["tuning_offsets"])
    assert list(kat["tuning_offsets"][:6]) == [8390.0, 5390.0, 7390.0, 6390.0, 6890.0, 6890.0]


def test_oracle_engine_is_the_images_bit_for_bit(kat, oracle):
    """every case: int16 audio, final scalars, and the stage taps where the fixture has them"""
    total = 0
    for name in case_names(kat):
        calls = calls_of(kat, name)
        tapped = (name + "_tap_in") in kat.files
        e = oracle.OracleEngine(taps=tapped)
        out = e.run(kat[name + "_iq"], calls)
        assert np.array_equal(out, kat[name + "_out"]), name
        assert np.array_equal(e.final().view(np.uint32), kat[name + "_final"].view(np.uint32)), (name, e.final(), kat[name + "_final"])
        total += len(out) // 128
        if tapped:
            for st in oracle.ENGINE_TAPS:
                key = f"{name}_tap_{st}"
                if key not in kat.files:
                    continue
                want = kat[key]
                got = np.stack(e.taps[st][:len(want)])[:, :want.shape[1]]
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), key
    assert total > 1500


def test_fixture_covers_what_the_sketch_can_ask_of_the_engine(kat):
    """every demodulator of the mode menu, every audio filter id, every AGC mode, the ALS filter in both outputs, the
    blanker, mute, the gain setters -- and the engine's behaviours a stand-in would miss: the AGC's hang, the PLL's lock,
    the wrap-around of the int16 store"""
    names = case_names(kat)
    calls = [c for n in names for c in calls_of(kat, n)]
    assert {c[2] for c in calls if c[1] == "setDemodMode"} >= {0, 1, 2, 3, 4, 5, 6}
    assert {c[2] for c in calls if c[1] == "setAudioFilter"} >= set(range(11))
    assert {c[2] for c in calls if c[1] == "setAGCmode"} >= {0, 1, 2, 3}
    assert {c[1] for c in calls} >= {"enableALSfilter", "setALSfilterNotch", "setALSfilterPeak", "setALSfilterAdaptive", "disableALSfilter",
                                     "enableNoiseBlanker", "setMute", "setInputGain", "setOutputGain", "setIQgainBalance", "enableAGC"}
    assert kat["sam_final"][6] == 1.0 and kat["sam_no_carrier_final"][6] == 0.0          # the PLL locks on a carrier and only then
    assert kat["blanker_on_final"][7] in (0.0, 1.0) and np.abs(kat["blanker_on_tap_nb"] - kat["blanker_on_tap_in"]).max() > 0
    # the hang AGC: after the 30 dB step down the output stays low for the hang time, then recovers
    for name, hang_blocks in (("agc_fast", 34), ("agc_medium", 172), ("agc_slow", 689)):
        rms = np.sqrt((kat[name + "_out"].astype(np.float64).reshape(-1, 128) ** 2).mean(1))
        held = rms[52:50 + hang_blocks - 2]
        assert held.max() < 0.2 * rms[45], name                                        # gain held where the loud signal left it
        assert rms[50 + hang_blocks + 4:].min() > held.mean() and rms[-1] > 1.4 * held.mean(), name   # and let go afterwards (slowly)
    w = kat["wrap_agc_off_out"].astype(np.int32)
    assert np.abs(np.diff(w)).max() > 40000                                            # a wrapped store, not a saturated one


# ---- GPU: the product against the image ---------------------------------------------------------------------------------
def _engine(rdsp, n_channels, max_blocks):
    from radiodsp_sdr_rx_amd.engine import Engine
    import oracle_lib
    return Engine(n_channels, max_blocks_per_call=max_blocks, tables=oracle_lib.engine_tables())


def _run_product(eng, iq_cases, calls, split):
    """iq_cases: int16 [n_channels, n, 2]; the engine's setters are called between calls at block boundaries"""
    import torch
    n = iq_cases.shape[1]
    nb = n // 128
    marks = sorted({0, nb} | {c[0] for c in calls} | set(range(0, nb, split)))
    out = np.zeros((iq_cases.shape[0], n), np.int16)
    for a, b in zip(marks[:-1], marks[1:]):
        for c in calls:
            if c[0] == a:
                r = getattr(eng, c[1])(*c[2:])
        d = torch.from_numpy(np.ascontiguousarray(iq_cases[:, a * 128:b * 128])).cuda()
        y = eng.update(d).cpu().numpy()
        assert np.array_equal(y[..., 0], y[..., 1])                                     # the same block on both outputs
        out[:, a * 128:b * 128] = y[..., 0]
    return out


@pytest.mark.gpu
def test_gpu_engine_tables_and_refusals(kat, rdsp):
    from radiodsp_sdr_rx_amd.engine import Engine
    from radiodsp_sdr_rx_amd._lib import RdspError
    import torch
    e = Engine(3, max_blocks_per_call=4)
    assert np.array_equal(e.sine_table(), kat["sine257"]) and np.array_equal(e.agc_curve()[:129], kat["agc_curve"][:129])
    assert [e.setDemodMode(m) for m in range(7)] == list(kat["tuning_offsets"])
    with pytest.raises(RdspError):                                                      # no tables, no audio
        e.update(torch.zeros((3, 128, 2), dtype=torch.int16, device="cuda"))


@pytest.mark.gpu
@pytest.mark.parametrize("split", [1, 7, 64])
def test_gpu_engine_is_the_images_bit_for_bit(kat, rdsp, split):
    """every case of the fixture through rdsp_engine_update, the stream cut into calls of `split` blocks (and at every
    setter of the case): int16 audio and the final scalars"""
    for name in case_names(kat):
        if split != 64 and len(kat[name + "_iq"]) > 300 * 128:
            continue                                                                   # the long AGC runs once
        eng = _engine(rdsp, 1, 64)
        eng.sketch_setup()
        out = _run_product(eng, kat[name + "_iq"][None], calls_of(kat, name), split)
        assert np.array_equal(out[0], kat[name + "_out"]), (name, split, int(np.argmax(out[0] != kat[name + "_out"])))
        got, want = eng.scalars()[0], kat[name + "_final"]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, got, want)
        eng.close()


@pytest.mark.gpu
def test_gpu_engine_many_channels_are_independent(kat, rdsp):
    """97 channels (a ragged last workgroup in every kernel) fed rotations of the sketch-default case: channel c must
    return what one engine alone returns for its input, which for the unrotated channels is the image's own audio"""
    iq = kat["lsb_sketch_iq"]
    n = len(iq)
    rolls = [0 if c % 5 == 0 else 128 * (c % 7) + c for c in range(97)]
    x = np.stack([np.roll(iq, r, axis=0) for r in rolls])
    eng = _engine(rdsp, 97, 16)
    eng.sketch_setup()
    out = _run_product(eng, x, [], 16)
    for c in range(97):
        if rolls[c] == 0:
            assert np.array_equal(out[c], kat["lsb_sketch_out"]), c
    import oracle_lib
    for c in (1, 33, 64, 96):
        assert np.array_equal(out[c], oracle_lib.OracleEngine().run(x[c])), c


@pytest.mark.gpu
@pytest.mark.parametrize("blanker", [False, True])
def test_gpu_engine_1100_channels_on_a_callers_stream(kat, rdsp, blanker):
    """1100 channels (35 front workgroups, 18 tail workgroups, the last ones ragged), two calls back to back on a
    non-default stream with the result read on that stream, with and without the blanker (whose rows past the last channel
    write to a spare slot) -- channel c must return what one engine alone returns"""
    import torch
    import oracle_lib
    name = "blanker_on" if blanker else "lsb_sketch"
    iq = kat[name + "_iq"][:24 * 128]
    nch = 1100
    rolls = [0 if c % 9 == 0 else 128 * (c % 5) + 3 * c for c in range(nch)]
    x = np.stack([np.roll(iq, r, axis=0) for r in rolls])
    eng = _engine(rdsp, nch, 12)
    eng.sketch_setup()
    if blanker:
        eng.enableNoiseBlanker()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        d = torch.from_numpy(x).cuda()
        out = torch.cat([eng.update(d[:, :12 * 128].contiguous(), stream=st.cuda_stream),
                         eng.update(d[:, 12 * 128:].contiguous(), stream=st.cuda_stream)], 1)[..., 0].cpu().numpy()
    for c in range(nch):
        if rolls[c] == 0:
            assert np.array_equal(out[c], kat[name + "_out"][:24 * 128]), c
    for c in (1, 319, 320, 639, 640, 959, 960, 1099):
        o = oracle_lib.OracleEngine()
        if blanker:
            o.call("enableNoiseBlanker")
        assert np.array_equal(out[c], o.run(x[c])), c


@pytest.mark.gpu
def test_gpu_engine_at_the_bench_shape(rdsp):
    """exactly `bench.py --config ENGINE`'s call (4096 receivers x 32 blocks of the synthetic generator's IQ, sketch set-up,
    two consecutive steps): 40 sampled channels, workgroup and wave boundaries among them, against the CPU restatement --
    which is the image's arithmetic (test_oracle_engine_is_the_images_bit_for_bit) -- bit for bit"""
    import torch
    import oracle_lib
    from radiodsp_sdr_rx_amd.chain import synth_iq
    nch, nblk = 4096, 32
    host = synth_iq(nch, 2 * nblk * 128, n_threads=8)
    eng = _engine(rdsp, nch, nblk)
    eng.sketch_setup()
    d = torch.from_numpy(host).cuda()
    out = torch.cat([eng.update(d[:, :nblk * 128].contiguous()), eng.update(d[:, nblk * 128:].contiguous())], 1)[..., 0].cpu().numpy()
    pick = sorted(set([0, 1, 7, 8, 9, 15, 16, 63, 64, 255, 256, 1023, 1024, 2047, 2048, 4088, 4095] + list(np.random.default_rng(1).integers(0, nch, 23))))
    for c in pick:
        assert np.array_equal(out[c], oracle_lib.OracleEngine().run(host[c])), c


@pytest.mark.gpu
def test_gpu_engine_reset_and_argument_checks(kat, rdsp):
    """rdsp_engine_reset gives the signal state of a fresh object and keeps the settings; calls the object cannot honour are
    refused with a message (too many blocks, short strides); zero blocks is a no-op"""
    import ctypes as C
    import torch
    from radiodsp_sdr_rx_amd._lib import RdspError
    iq = kat["als_notch_iq"][:16 * 128]
    calls = calls_of(kat, "als_notch")
    eng = _engine(rdsp, 2, 16)
    eng.sketch_setup()
    for c in calls:
        getattr(eng, c[1])(*c[2:])
    d = torch.from_numpy(np.stack([iq, iq])).cuda()
    first = eng.update(d).cpu().numpy()
    assert np.array_equal(first[0, :, 0], kat["als_notch_out"][:16 * 128])
    eng.update(d)                                                    # move every state on, then back to the start
    eng.reset()
    assert np.array_equal(eng.update(d).cpu().numpy(), first)        # filter states, lines, rings, AGC, the ALS taps: all as constructed
    with pytest.raises(RdspError):
        eng.update(torch.zeros((2, 17 * 128, 2), dtype=torch.int16, device="cuda"))
    lib = eng.lib
    assert lib.rdsp_engine_update(eng.h, d.data_ptr(), 16 * 128, 0, d.data_ptr(), 16 * 128, None) == 0
    assert lib.rdsp_engine_update(eng.h, d.data_ptr(), 100, 16, d.data_ptr(), 16 * 128, None) == -1
    assert b"n_blocks" in lib.rdsp_last_error()
    assert lib.rdsp_engine_update(None, d.data_ptr(), 16 * 128, 1, d.data_ptr(), 16 * 128, None) == -1


# ---- random control sessions ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sessions():
    return np.load(os.path.join(HERE, "golden", "engine_sessions.npz"))


def test_oracle_engine_random_sessions_are_the_images(sessions, oracle):
    """tests/golden/engine_sessions.npz: six drawn sequences of the sketch's setter calls over changing signals, run by the
    image's AudioSDR::update() (tests/golden/make_engine_sessions.py): the restatement's audio and final scalars, bit for bit"""
    seen = set()
    for name in case_names(sessions):
        calls = calls_of(sessions, name)
        seen |= {c[1] for c in calls}
        e = oracle.OracleEngine()
        assert np.array_equal(e.run(sessions[name + "_iq"], calls), sessions[name + "_out"]), name
        assert np.array_equal(e.final().view(np.uint32), sessions[name + "_final"].view(np.uint32)), name
    assert len(seen) >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("split", [1, 5])
def test_gpu_engine_random_sessions_are_the_images(sessions, rdsp, split):
    for name in case_names(sessions):
        eng = _engine(rdsp, 1, 8)
        eng.sketch_setup()
        out = _run_product(eng, sessions[name + "_iq"][None], calls_of(sessions, name), split)
        assert np.array_equal(out[0], sessions[name + "_out"]), (name, split, int(np.argmax(out[0] != sessions[name + "_out"])))
        assert np.array_equal(eng.scalars()[0].view(np.uint32), sessions[name + "_final"].view(np.uint32)), name
        eng.close()


def _drawn_session(seed, n_blocks):
    r = np.random.default_rng(seed)
    t = np.arange(n_blocks * 128)
    z = np.zeros(len(t), np.complex128)
    for _ in range(int(r.integers(1, 5))):
        z += r.uniform(0.02, 0.3) * np.exp(2j * np.pi * r.uniform(4000, 10000) / 44100.0 * t + 1j * r.uniform(0, 6))
    z += r.uniform(0, 0.3) * np.exp(2j * np.pi * 6890.0 / 44100.0 * t) * (1 + 0.5 * np.sin(2 * np.pi * 500.0 / 44100.0 * t))
    z += r.uniform(0, 0.05) * (r.standard_normal(len(t)) + 1j * r.standard_normal(len(t)))
    z *= np.repeat(r.choice([0.02, 0.2, 1.0, 3.0], n_blocks // 4 + 1), 4 * 128)[:len(t)]
    iq = np.stack([np.clip(np.round(z.real * 32767), -32768, 32767), np.clip(np.round(z.imag * 32767), -32768, 32767)], 1).astype(np.int16)
    menu = [["setDemodMode", int(m)] for m in range(7)] + [["setAudioFilter", int(f)] for f in range(11)] + [["setAGCmode", int(a)] for a in range(4)]
    menu += [["enableAGC"], ["enableALSfilter"], ["disableALSfilter"], ["setALSfilterNotch"], ["setALSfilterPeak"], ["setALSfilterAdaptive"],
             ["enableNoiseBlanker"], ["disableNoiseBlanker"], ["setMute", 1], ["setMute", 0], ["enableAudioFilter"]]
    menu += [["setInputGain", g] for g in (0.25, 1.0, 3.0)] + [["setOutputGain", g] for g in (0.2, 0.5, 0.9)] + [["setIQgainBalance", g] for g in (0.95, 1.0, 1.02)]
    calls = [[int(b)] + menu[int(r.integers(0, len(menu)))] for b in sorted(r.integers(0, n_blocks, n_blocks // 3))]
    return iq, calls


@pytest.mark.gpu
def test_gpu_engine_drawn_sessions_against_the_restatement(rdsp):
    """twenty more sessions of 60 blocks, drawn here, five receivers each with a session of its own setter calls applied to
    ALL of them (the object's settings are the object's): rdsp_engine_t against the CPU restatement, bit for bit, calls cut
    at every setter and every 1 ... 7 blocks"""
    import oracle_lib
    for s in range(20):
        sess = [_drawn_session(100 * s + c, 60) for c in range(5)]
        calls = sess[0][1]
        x = np.stack([q for q, _ in sess])
        eng = _engine(rdsp, 5, 8)
        eng.sketch_setup()
        out = _run_product(eng, x, calls, 1 + s % 7)
        for c in range(5):
            assert np.array_equal(out[c], oracle_lib.OracleEngine().run(x[c], calls)), (s, c)
        eng.close()


@pytest.mark.gpu
def test_gpu_engine_receiver_groups(kat, sessions, rdsp):
    """rdsp_engine_set_groups: one object, groups of consecutive channels each in its own mode with its own filter, AGC and
    gains, driven through their own setter sequences in the same calls -- five fixture cases side by side in 23 channels
    (groups of 3, 9, 1, 8 and 2 channels: none aligned to a workgroup), every channel against the image's audio for its
    group's case; then a regrouping in mid-stream"""
    names = ["lsb_sketch", "am", "session2", "als_notch", "menu_walk"]
    fx = {n: (kat if n + "_iq" in kat.files else sessions) for n in names}
    nb = min(len(fx[n][n + "_iq"]) // 128 for n in names)
    first = [0, 3, 12, 13, 21]
    size = [3, 9, 1, 8, 2]
    x = np.concatenate([np.repeat(fx[n][n + "_iq"][None, :nb * 128], k, 0) for n, k in zip(names, size)])
    eng = _engine(rdsp, 23, 8)
    eng.sketch_setup()
    eng.set_groups(first)
    calls = {n: calls_of(fx[n], n) for n in names}
    import torch
    out = np.zeros((23, nb * 128), np.int16)
    marks = sorted({0, nb} | {c[0] for n in names for c in calls[n] if c[0] < nb} | set(range(0, nb, 5)))
    for a, b in zip(marks[:-1], marks[1:]):
        for g, n in enumerate(names):
            eng.select_group(g)
            for c in calls[n]:
                if c[0] == a:
                    getattr(eng, c[1])(*c[2:])
        y = eng.update(torch.from_numpy(np.ascontiguousarray(x[:, a * 128:b * 128])).cuda()).cpu().numpy()
        out[:, a * 128:b * 128] = y[..., 0]
    for g, n in enumerate(names):
        for c in range(first[g], first[g] + size[g]):
            assert np.array_equal(out[c], fx[n][n + "_out"][:nb * 128]), (n, c)
    # regrouping keeps every channel's signal state; new groups start as copies of the group their first channel was in
    eng.select_group(-1)
    assert eng.lib.rdsp_engine_groups(eng.h) == 5
    eng.set_groups([0, 12])
    assert eng.lib.rdsp_engine_groups(eng.h) == 2
    from radiodsp_sdr_rx_amd._lib import RdspError
    with pytest.raises(RdspError):
        eng.set_groups([0, 30])
    with pytest.raises(RdspError):
        eng.select_group(2)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["lsb_sketch", "als_notch", "blanker_on", "sam", "menu_walk"])
def test_gpu_engine_state_as_data(kat, rdsp, name):
    """rdsp_engine_save_state / load_state: a receiver stopped after k blocks and continued in ANOTHER object (other channel
    count, other ring size, the receiver at another channel index) plays the image's audio to the end, bit for bit"""
    import torch
    iq, calls, want = kat[name + "_iq"], calls_of(kat, name), kat[name + "_out"]
    nb = len(iq) // 128
    k = nb // 2 - (nb // 2) % 2
    a = _engine(rdsp, 3, 8)
    a.sketch_setup()
    x = np.stack([iq[::-1].copy(), iq, iq])
    first = _run_product(a, x[:, :k * 128], [c for c in calls if c[0] < k], 3)
    assert np.array_equal(first[1], want[:k * 128])
    blob = a.save_state(1, 1)
    b = _engine(rdsp, 6, 24)
    b.sketch_setup()
    for c in calls:                                                 # the settings are not in the blob: the host makes the same calls
        if c[0] < k:
            getattr(b, c[1])(*c[2:])
    b.update(torch.from_numpy(np.zeros((6, 128, 2), np.int16)).cuda())   # b has a past of its own (its rings sit elsewhere)
    b.load_state(4, blob)
    y = np.zeros((6, (nb - k) * 128, 2), np.int16)
    y[4] = iq[k * 128:]
    rest = _run_product(b, y, [[c[0] - k] + c[1:] for c in calls if c[0] >= k], 7)
    assert np.array_equal(rest[4], want[k * 128:]), name
    from radiodsp_sdr_rx_amd._lib import RdspError
    with pytest.raises(RdspError):
        b.load_state(0, np.zeros(64, np.uint8))                      # not a blob
    with pytest.raises(RdspError):
        b.load_state(6, blob)                                        # no such channel
