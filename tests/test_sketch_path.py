"""The sketch's whole audio path against the reference's own compiled code, end to end.

RadioDSP_SDR_RX.ino:71-89 wires  IQ input -> preProcessor -> SDR -> record queues, loop() (:198) runs
doConvolutionalProcessing between those and the play queues.  tests/golden/sketch_kat.npz holds what the firmware image's
AudioSDRpreProcessor::update, AudioSDR::update and doConvolutionalProcessing make of seeded int16 IQ when chained block
by block at the sketch's start-up settings (tests/golden/make_sketch_kat.py, build container; `--check` reproduces it):
the blocks after the pre-processor, after the engine, and the audio that is played.

`-m "not gpu"`: the oracle's three restatements chained the same way.  `-m gpu`: the product -- rdsp_preproc_t,
rdsp_engine_t and the CONV-stage chain through the C-ABI, first as three calls on a whole stream, then as the sketch's
graph (nodes, connections, queues, the loop() body) ticked block by block.  The pre-processor's and the engine's int16
blocks must be the image's bit for bit; the played audio within one count (the CONV stage's transform is the one place
where this build's arithmetic is not the image's operation for operation: tests/test_firmware_kat.py, 1.5e-6)."""
import os

import numpy as np
import pytest

from cases import CONV_LITERAL

HERE = os.path.dirname(os.path.abspath(__file__))
FULL = ("sketch_path", "sketch_path_slip")
PRE_ONLY = ("pre_slip_i", "pre_slip_q", "pre_clean", "pre_noise", "pre_swap")
CONV = dict(CONV_LITERAL, lms_nr=15)                           # INO:172-183


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(HERE, "golden", "sketch_kat.npz"))


def one_count(a, b):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32))
    return d.max() <= 1 and (d > 0).mean() < 0.02


def test_fixture_shows_the_preprocessor_at_work(kat):
    """a late Q rail is found at the eleventh block and repaired by delaying I; a late I rail first gets the wrong remedy
    (I later still), then the right one (Q); clean and carrier-less streams pass untouched; swapIQ swaps"""
    s = kat["pre_slip_q_pre_state"]
    assert list(s[9]) == [0, 10, 10, 1] and list(s[10]) == [1, 0, 1, 1] and s[-1, 0] == 1 and s[-1, 1] == 0
    assert np.array_equal(kat["pre_slip_q_pre"][12 * 128 + 1:, 0], kat["pre_slip_q_iq"][12 * 128:-1, 0])
    s = kat["pre_slip_i_pre_state"]
    assert list(s[:, 0][[9, 10, 20, 21, 39]]) == [0, 1, 1, -1, -1]
    for name in ("pre_clean", "pre_noise"):
        assert np.array_equal(kat[name + "_pre"], kat[name + "_iq"]) and not kat[name + "_pre_state"][:, 0].any()
    assert kat["pre_noise_pre_state"][-1, 2] == 0 and kat["pre_clean_pre_state"][-1, 2] == 20
    assert np.array_equal(kat["pre_swap_pre"], kat["pre_swap_iq"][:, ::-1])
    assert kat["sketch_path_slip_pre_state"][-1, 0] == 1 and not kat["sketch_path_pre_state"][:, 0].any()


def test_oracle_preprocessor_is_the_images(kat, oracle):
    for name in FULL + PRE_ONLY:
        out, st = oracle.OraclePreProcessor(swap=name == "pre_swap").run(kat[name + "_iq"])
        assert np.array_equal(st, kat[name + "_pre_state"]), name
        assert np.array_equal(out, kat[name + "_pre"]), name


def test_oracle_chain_of_the_three_stages_is_the_images(kat, oracle):
    for name in FULL:
        pre, _ = oracle.OraclePreProcessor().run(kat[name + "_iq"])
        sdr = oracle.OracleEngine().run(pre)
        assert np.array_equal(sdr, kat[name + "_sdr"]), name
        o16, _ = oracle.OracleChain(**CONV).process(np.stack([sdr, sdr], 1))
        assert o16.shape == kat[name + "_audio"].shape and one_count(o16, kat[name + "_audio"]), name


# ---- GPU ----------------------------------------------------------------------------------------------------------------
def _objects(nch, max_blocks):
    import oracle_lib
    from radiodsp_sdr_rx_amd.chain import Chain
    from radiodsp_sdr_rx_amd.engine import Engine, PreProcessor
    pre = PreProcessor(nch)
    pre.startAutoI2SerrorDetection()                              # INO:117
    eng = Engine(nch, max_blocks_per_call=max_blocks, tables=oracle_lib.engine_tables())
    assert eng.sketch_setup() == 8390.0                           # INO:120-139: TuningOffset
    return pre, eng, Chain(nch, max_blocks_per_call=max_blocks, **CONV)


@pytest.mark.gpu
def test_gpu_preprocessor_is_the_images(kat, rdsp):
    import torch
    from radiodsp_sdr_rx_amd.engine import PreProcessor
    for name in FULL + PRE_ONLY:
        iq = kat[name + "_iq"]
        for split in (1, 5, len(iq) // 128):
            p = PreProcessor(2)
            p.startAutoI2SerrorDetection()
            p.swapIQ(name == "pre_swap")
            x = torch.from_numpy(np.stack([iq, iq[::-1].copy()])).cuda()           # a second channel with another stream beside it
            out = torch.cat([p.update(x[:, a:a + split * 128].contiguous()) for a in range(0, len(iq), split * 128)], 1)
            assert np.array_equal(out[0].cpu().numpy(), kat[name + "_pre"]), (name, split)
            assert list(p.state()[0]) == list(kat[name + "_pre_state"][-1]), (name, split)


@pytest.mark.gpu
def test_gpu_three_stages_on_a_stream(kat, rdsp):
    import torch
    for name in FULL:
        iq = kat[name + "_iq"]
        pre, eng, conv = _objects(3, 32)
        x = torch.from_numpy(np.stack([iq, iq, iq])).cuda()
        audio = []
        for a in range(0, len(iq), 32 * 128):
            p = pre.update(x[:, a:a + 32 * 128].contiguous())
            if a == 0:
                assert np.array_equal(p[1].cpu().numpy(), kat[name + "_pre"][:32 * 128])
            s = eng.update(p)
            assert np.array_equal(s[2, :, 0].cpu().numpy(), kat[name + "_sdr"][a:a + 32 * 128]), (name, a)
            audio.append(conv.process(s))
        got = torch.cat(audio, 1).cpu().numpy()
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])
        assert one_count(got[0], kat[name + "_audio"]), name


@pytest.mark.gpu
def test_gpu_the_sketchs_graph_block_by_block(kat, rdsp):
    """INO:52-67 objects, INO:71-89 connections (the display's analysers aside: tests/test_graph.py has those), INO:198's
    loop(): `if both record queues have a block: doConvolutionalProcessing`"""
    import torch
    from radiodsp_sdr_rx_amd.graph import Graph
    name = "sketch_path_slip"
    iq = kat[name + "_iq"]
    nblk = len(iq) // 128
    pre, eng, conv = _objects(1, 2)
    unit = conv.call_unit_blocks
    g = Graph(1)
    g.AudioMemory(40)                                             # INO:151
    IQinput = g.input_node()                                      # INO:52
    preProcessor = g.preproc_node(pre)                            # INO:53
    SDR = g.engine_node(eng)                                      # INO:54
    Q_in_L, Q_in_R = g.record_queue(), g.record_queue()           # INO:64-65
    Q_out_L, Q_out_R = g.play_queue(), g.play_queue()             # INO:66-67
    heard, seen = [], []

    def out_update(n):                                            # AudioOutputI2S audio_out, INO:55
        l, r = n.receiveReadOnly(0), n.receiveReadOnly(1)
        if l is not None and r is not None:
            heard.append(np.stack([l.data()[0].copy(), r.data()[0].copy()], axis=1))
        n.release(l); n.release(r)

    audio_out = g.node(2, out_update)
    g.AudioConnection(IQinput, 0, preProcessor, 0)                # c1
    g.AudioConnection(IQinput, 1, preProcessor, 1)                # c2
    g.AudioConnection(preProcessor, 0, SDR, 0)                    # a3
    g.AudioConnection(preProcessor, 1, SDR, 1)                    # a4
    g.AudioConnection(SDR, 0, Q_in_L, 0)                          # c5
    g.AudioConnection(SDR, 1, Q_in_R, 0)                          # c6
    g.AudioConnection(Q_out_L, 0, audio_out, 0)                   # c7
    g.AudioConnection(Q_out_R, 0, audio_out, 1)                   # c8
    Q_in_L.begin(); Q_in_R.begin()                                # CONV:205-206
    for b in range(nblk + 2):
        if b < nblk:
            IQinput.push(iq[None, b * 128:(b + 1) * 128, 0], iq[None, b * 128:(b + 1) * 128, 1])
        assert g.update_all() == 0
        while Q_in_L.available() >= unit and Q_in_R.available() >= unit:  # loop(), INO:198: the chain's call unit at a time
            ls, rs = [], []
            for _ in range(unit):
                ls.append(Q_in_L.readBuffer().copy()); rs.append(Q_in_R.readBuffer().copy())
                Q_in_L.freeBuffer(); Q_in_R.freeBuffer()
                seen.append(ls[-1][0])
            x = np.stack([np.concatenate(ls, 1), np.concatenate(rs, 1)], 2)
            y = conv.doConvolutionalProcessing(15.0, True, 300.0, 4000.0, torch.from_numpy(x).cuda()).cpu().numpy()
            for k in range(unit):
                ol, orr = Q_out_L.getBuffer(), Q_out_R.getBuffer()
                ol[:] = y[:, k * 128:(k + 1) * 128, 0]; orr[:] = y[:, k * 128:(k + 1) * 128, 1]
                assert Q_out_L.playBuffer() == 0 and Q_out_R.playBuffer() == 0
                assert g.update_all() == 0
    assert SDR.status() == 0 and preProcessor.status() == 0
    assert np.array_equal(np.concatenate(seen), kat[name + "_sdr"])
    assert one_count(np.concatenate(heard), kat[name + "_audio"])
    assert list(pre.state()[0]) == list(kat[name + "_pre_state"][-1])


@pytest.mark.gpu
def test_gpu_one_chain_as_the_sketch_as_shipped(kat, rdsp):
    """rdsp_sdr_set_engine_literal(chain, 1): ONE rdsp_chain_t, created as the bare CONV stage, with the reference's own
    pre-processor and engine in front of it.  The sketch's set-up through the chain's rdsp_sdr_* / rdsp_pre_* entry points
    with the chain's enums (rdsp_demod_t, rdsp_audio_filter_t: INO:117-139 as tests/host/rdsp_binding.h spells it) reaches
    those objects; rdsp_chain_process and the graph's SDR node then play the image's audio to one count -- and a mode-menu
    walk through the same entry points gives the engine fixture's audio, engine and CONV stage in one call."""
    import torch
    import oracle_lib
    from radiodsp_sdr_rx_amd.chain import Chain
    from radiodsp_sdr_rx_amd.config import AUDIO_FILTER, DEMOD
    from radiodsp_sdr_rx_amd._lib import RdspError
    name = "sketch_path_slip"
    iq = kat[name + "_iq"]
    ch = Chain(2, max_blocks_per_call=8, **CONV)
    ch.set_engine_literal(True, oracle_lib.engine_tables())
    ch.startAutoI2SerrorDetection()                                # INO:117
    ch.enableAGC(); ch.setAGCmode(2); ch.disableALSfilter(); ch.disableNoiseBlanker()   # INO:120-131
    ch.setInputGain(1.0); ch.setOutputGain(0.5); ch.setIQgainBalance(1.020)             # INO:133-135
    ch.enableAudioFilter(); ch.setAudioFilter(AUDIO_FILTER["audio2700"])                 # INO:137-138
    assert ch.lib.rdsp_sdr_setDemodMode(ch.h, DEMOD["LSB"], None) == 8390                # INO:139: TuningOffset
    ch.setMute(False)                                              # INO:177
    x = torch.from_numpy(np.stack([iq, iq])).cuda()
    got = torch.cat([ch.process(x[:, a:a + 8 * 128].contiguous()) for a in range(0, len(iq), 8 * 128)], 1).cpu().numpy()
    assert np.array_equal(got[0], got[1]) and one_count(got[0], kat[name + "_audio"])
    with pytest.raises(RdspError):                                 # a chain that is more than the CONV stage is refused
        Chain(1, max_blocks_per_call=8, **dict(CONV, agc_mode="fast")).set_engine_literal(True)
    # the mode menu through the chain's entry points: engine_kat's menu_walk, CONV stage switched to pass-through checks
    ek = np.load(os.path.join(HERE, "golden", "engine_kat.npz"))
    import json
    calls = json.loads(str(ek["menu_walk_calls"]))
    c2 = Chain(1, max_blocks_per_call=2, **dict(CONV, lms_nr=0))
    c2.set_engine_literal(True, oracle_lib.engine_tables())
    for cfun, arg in (("enableAGC", None), ("setAGCmode", 2), ("disableALSfilter", None), ("disableNoiseBlanker", None), ("setInputGain", 1.0),
                      ("setOutputGain", 0.5), ("setIQgainBalance", 1.02), ("enableAudioFilter", None)):
        getattr(c2, cfun)(*([] if arg is None else [arg]))
    c2.setAudioFilter(AUDIO_FILTER["audio2700"]); c2.lib.rdsp_sdr_setDemodMode(c2.h, DEMOD["LSB"], None)
    eng_mode = {0: "LSB", 1: "USB", 2: "CW_LSB", 3: "CW_USB", 4: "AM", 5: "SAM"}
    eng_filt = {0: "audioAM", 1: "audioCW", 3: "audio2100", 6: "audio2700", 8: "audio3100"}
    miq = ek["menu_walk_iq"]
    import radiodsp_sdr_rx_amd.engine as E
    ref_eng = E.Engine(1, max_blocks_per_call=2, tables=oracle_lib.engine_tables())
    ref_eng.sketch_setup()
    for b in range(0, 36, 2):                                      # the CONV stage takes two blocks a call; up to block 36 the fixture's menu calls sit on even blocks
        for c in calls:
            if c[0] in (b, b + 1):
                assert c[0] == b
                if c[1] == "setDemodMode":
                    c2.lib.rdsp_sdr_setDemodMode(c2.h, DEMOD[eng_mode[c[2]]], None); ref_eng.setDemodMode(c[2])
                elif c[1] == "setAudioFilter":
                    c2.setAudioFilter(AUDIO_FILTER[eng_filt[c[2]]]); ref_eng.setAudioFilter(c[2])
                else:
                    getattr(c2, c[1])(*c[2:]); getattr(ref_eng, c[1])(*c[2:])
        d = torch.from_numpy(miq[None, b * 128:(b + 2) * 128].copy()).cuda()
        y = c2.process(d)
        s = ref_eng.update(d)                                       # the engine alone, then a CONV stage of its own
        assert np.array_equal(s[0, :, 0].cpu().numpy(), ek["menu_walk_out"][b * 128:(b + 2) * 128])
        if b == 0:
            conv_ref = Chain(1, max_blocks_per_call=2, **dict(CONV, lms_nr=0))
        assert torch.equal(y, conv_ref.process(s))
