"""Block-graph runtime (csrc/rdsp_graph.c): the AudioStream semantics the reference
relies on, tested on the CPU with pure-plumbing nodes; the SDR engine node that runs
the GPU chain is tested under -m gpu with the K1 configuration (1 channel, 96 kHz IQ,
128-sample blocks, USB, NR/notch off) wired like RadioDSP_SDR_RX.ino:71-89."""
import numpy as np
import pytest

from cases import K1


def test_wiring_fanout_refcounts_and_update_order(rdsp):
    from radiodsp_sdr_rx_amd.graph import Graph
    g = Graph(n_channels=2)
    g.AudioMemory(8)
    order = []
    src = g.input_node()                      # IQinput

    def pre_update(n):                        # preProcessor: pass-through, 2 in / 2 out
        order.append("pre")
        bi, bq = n.receiveReadOnly(0), n.receiveReadOnly(1)
        if bi is None or bq is None:          # FFTIQ.cpp:72: silent early return
            n.release(bi); n.release(bq)
            return
        n.transmit(bi, 0); n.transmit(bq, 1)
        n.release(bi); n.release(bq)

    pre = g.node(2, pre_update)
    seen = {}

    def tap_update(n):                        # second consumer of IQinput port 0 (fan-out, INO:71,75)
        order.append("tap")
        b = n.receiveReadOnly(0)
        if b is not None:
            seen["ref"] = b.refcount()
            seen["val"] = int(b.data()[1, 5])
            n.release(b)

    tap = g.node(1, tap_update)
    ql, qr = g.record_queue(), g.record_queue()
    g.AudioConnection(src, 0, pre, 0)
    g.AudioConnection(src, 1, pre, 1)
    g.AudioConnection(src, 0, tap, 0)         # fan-out of port 0
    g.AudioConnection(pre, 0, ql, 0)
    g.AudioConnection(pre, 1, qr, 0)
    ql.begin(); qr.begin()
    for t in range(3):
        i = np.full((2, 128), 10 * t, np.int16); i[1, 5] = 77 + t
        q = np.full((2, 128), -t, np.int16)
        src.push(i, q)
        assert g.update_all() == 0
    assert order[:2] == ["pre", "tap"]        # creation order
    assert seen["val"] == 79
    assert ql.available() == 3 and qr.available() == 3
    b0 = ql.readBuffer()
    assert b0[0, 0] == 0 and b0[1, 5] == 77
    assert ql.readBuffer() is None            # one user block at a time until freeBuffer
    ql.freeBuffer()
    assert ql.readBuffer()[0, 0] == 10
    ql.freeBuffer()
    assert qr.readBuffer()[0, 0] == 0
    qr.freeBuffer()
    used, peak = g.memory_usage()
    assert used == 3 and peak <= 8            # 1 left in ql, 2 in qr
    # a tick with no capture: nodes see no input and return early
    assert g.update_all() == 0 and ql.available() == 1


def test_pool_exhaustion_and_queue_gate(rdsp):
    from radiodsp_sdr_rx_amd.graph import Graph
    g = Graph(1)
    g.AudioMemory(4)
    src = g.input_node()
    q = g.record_queue()
    g.AudioConnection(src, 0, q, 0)
    q.begin()
    z = np.zeros((1, 128), np.int16)
    for _ in range(6):
        src.push(z, z)
        g.update_all()
    # 4 blocks in the pool; each tick needs 2 (I and Q): after two ticks the pool holds
    # 2 queued I blocks and the Q blocks were freed; later allocations succeed until empty
    assert q.available() <= 4 and g.memory_usage()[0] <= 4
    # `available() > N_BLOCKS` (CONV:231) is a strict gate: N_BLOCKS+1 must be queued
    n_blocks = 1
    assert (q.available() > n_blocks) == (q.available() >= 2)


def test_play_queue_and_interrupt_gate(rdsp):
    from radiodsp_sdr_rx_amd.graph import Graph
    g = Graph(1)
    g.AudioMemory(6)
    pq = g.play_queue()                       # Q_out_L
    got = []

    def sink_update(n):                       # audio_out
        b = n.receiveReadOnly(0)
        if b is not None:
            got.append(b.data().copy())
            n.release(b)

    sink = g.node(1, sink_update)
    fft = g.node(1, sink_update)              # AudioFFT also listens to Q_out_L (INO:87-88)
    g.AudioConnection(pq, 0, fft, 0)
    g.AudioConnection(pq, 0, sink, 0)
    buf = pq.getBuffer()
    buf[:] = 123
    assert pq.playBuffer() == 0
    g.AudioNoInterrupts()                     # CONV:211: the ISR is held off
    assert g.update_all() != 0 and not got
    g.AudioInterrupts()
    assert g.update_all() == 0
    assert len(got) == 2 and (got[0] == 123).all()
    assert g.memory_usage()[0] == 0


@pytest.mark.gpu
def test_engine_node_with_a_pipelined_chain_is_bit_identical_to_resident_processing(rdsp):
    """A chain in pipelined mode finishes its output on an internal stream; the engine node
    flushes before it copies the audio back (3 channels, K3: NLMS tail stage + AGC)."""
    import torch
    from cases import K3
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.graph import Graph
    nch, nblk = 3, 48
    iq = synth_iq(nch, nblk * 128)
    ref = Chain(nch, max_blocks_per_call=8, **K3)
    want = np.concatenate([ref.process(torch.from_numpy(np.ascontiguousarray(iq[:, k * 1024:(k + 1) * 1024])).cuda()).cpu().numpy()
                           for k in range(nblk // 8)], axis=1)
    chain = Chain(nch, max_blocks_per_call=8, **K3)
    chain.set_pipelined(True)
    g = Graph(nch)
    g.AudioMemory(60)
    src, sdr = g.input_node(), g.sdr_node(chain)
    ql, qr = g.record_queue(), g.record_queue()
    g.AudioConnection(src, 0, sdr, 0); g.AudioConnection(src, 1, sdr, 1)
    g.AudioConnection(sdr, 0, ql, 0); g.AudioConnection(sdr, 1, qr, 0)
    ql.begin(); qr.begin()
    L, R = [], []
    for b in range(nblk + 8):
        if b < nblk:
            blk = np.ascontiguousarray(iq[:, b * 128:(b + 1) * 128])
            src.push(np.ascontiguousarray(blk[..., 0]), np.ascontiguousarray(blk[..., 1]))
        assert g.update_all() == 0
        while ql.available() > 0 and qr.available() > 0:
            L.append(ql.readBuffer().copy()); ql.freeBuffer()
            R.append(qr.readBuffer().copy()); qr.freeBuffer()
    assert sdr.status() == 0
    got = np.stack([np.concatenate(L, axis=1), np.concatenate(R, axis=1)], axis=2)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_k1_one_channel_through_the_graph_matches_oracle(rdsp, oracle):
    """BASELINE config K1 through the update()/connect() API with the GPU engine node."""
    import torch
    assert torch.cuda.is_available()
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.graph import Graph
    nblk = 64
    iq = synth_iq(1, nblk * 128)
    chain = Chain(1, max_blocks_per_call=8, **K1)
    g = Graph(1)
    g.AudioMemory(40)                          # INO:151
    IQinput = g.input_node()                   # INO:52

    def pre_update(n):                         # AudioSDRpreProcessor: pass-through here
        bi, bq = n.receiveReadOnly(0), n.receiveReadOnly(1)
        if bi is not None and bq is not None:
            n.transmit(bi, 0); n.transmit(bq, 1)
        n.release(bi); n.release(bq)

    preProcessor = g.node(2, pre_update)       # INO:53
    SDR = g.sdr_node(chain)                    # INO:54
    Q_in_L, Q_in_R = g.record_queue(), g.record_queue()   # INO:63-64
    g.AudioConnection(IQinput, 0, preProcessor, 0)        # c1
    g.AudioConnection(IQinput, 1, preProcessor, 1)        # c2
    g.AudioConnection(preProcessor, 0, SDR, 0)            # a3
    g.AudioConnection(preProcessor, 1, SDR, 1)            # a4
    g.AudioConnection(SDR, 0, Q_in_L, 0)                  # c5
    g.AudioConnection(SDR, 1, Q_in_R, 0)                  # c6
    Q_in_L.begin(); Q_in_R.begin()                        # CONV:205-206
    L, R = [], []
    for b in range(nblk + 8):                  # a few extra ticks drain the engine's output fifo
        if b < nblk:
            blk = iq[0, b * 128:(b + 1) * 128]
            IQinput.push(blk[None, :, 0], blk[None, :, 1])
        assert g.update_all() == 0
        while Q_in_L.available() > 0 and Q_in_R.available() > 0:
            L.append(Q_in_L.readBuffer()[0].copy()); Q_in_L.freeBuffer()
            R.append(Q_in_R.readBuffer()[0].copy()); Q_in_R.freeBuffer()
    assert SDR.status() == 0
    L, R = np.concatenate(L), np.concatenate(R)
    r16, _ = oracle.OracleChain(**K1).process(iq[0])
    assert len(L) == len(r16) == nblk * 128 // 4
    d = np.abs(np.stack([L, R], 1).astype(np.int32) - r16.astype(np.int32))
    assert d.max() <= 1
    assert g.memory_usage()[1] <= 40


@pytest.mark.gpu
def test_the_whole_graph_of_the_sketch(rdsp, oracle):
    """Every AudioConnection of RadioDSP_SDR_RX.ino:71-89 and the set-up calls of :144-156:
    IQinput -> preProcessor -> SDR -> Q_in_L/R -(loop)-> Q_out_L/R -> audio_out, the panadapter
    branch IQinput -> biquad1/2 -> FFT, and AudioFFT on Q_out_L.  The audio that reaches audio_out
    is the chain's (oracle, +-1 LSB); both analysers produce spectra from what flowed past them."""
    import torch
    from cases import K1
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
    from radiodsp_sdr_rx_amd.filters import AnalyzeFFT1024, FilterBiquad
    from radiodsp_sdr_rx_amd.graph import Graph
    from radiodsp_sdr_rx_amd.spectrum import AnalyzeFFT256IQ
    assert torch.cuda.is_available()
    nblk = 96
    iq = synth_iq(1, nblk * 128)
    chain = Chain(1, max_blocks_per_call=8, **K1)
    g = Graph(1)
    g.AudioMemory(40)                                            # INO:151
    IQinput = g.input_node()                                     # INO:52

    def pre_update(n):                                           # AudioSDRpreProcessor: pass-through
        bi, bq = n.receiveReadOnly(0), n.receiveReadOnly(1)
        if bi is not None and bq is not None:
            n.transmit(bi, 0); n.transmit(bq, 1)
        n.release(bi); n.release(bq)

    preProcessor = g.node(2, pre_update)                         # INO:53
    SDR = g.sdr_node(chain)                                      # INO:54
    fft = AnalyzeFFT256IQ(1, naverage=30, window="AudioWindowHanning256")   # INO:57,144-145
    FFT = g.spectrum_node(fft)
    afft = AnalyzeFFT1024(1, window="AudioWindowHanning1024")    # INO:58,147
    afft.averageTogether(30)                                     # INO:148
    AudioFFT = g.fft1024_node(afft)
    b1, b2 = FilterBiquad(1), FilterBiquad(1)                    # INO:59-60
    b1.setHighpass(0, 500, 0.5); b2.setHighpass(0, 500, 0.5)     # INO:155-156
    biquad1, biquad2 = g.biquad_node(b1), g.biquad_node(b2)
    Q_in_L, Q_in_R = g.record_queue(), g.record_queue()          # INO:64-65
    Q_out_L, Q_out_R = g.play_queue(), g.play_queue()            # INO:66-67
    heard = []

    def out_update(n):                                           # AudioOutputI2S audio_out
        l, r = n.receiveReadOnly(0), n.receiveReadOnly(1)
        if l is not None and r is not None:
            heard.append(np.stack([l.data()[0].copy(), r.data()[0].copy()], axis=1))
        n.release(l); n.release(r)

    audio_out = g.node(2, out_update)                            # INO:55
    g.AudioConnection(IQinput, 0, preProcessor, 0)               # c1
    g.AudioConnection(IQinput, 1, preProcessor, 1)               # c2
    g.AudioConnection(IQinput, 0, biquad1, 0)                    # c2f1
    g.AudioConnection(IQinput, 1, biquad2, 0)                    # c2f2
    g.AudioConnection(biquad1, 0, FFT, 0)                        # c2f11
    g.AudioConnection(biquad2, 0, FFT, 1)                        # c2f22
    g.AudioConnection(preProcessor, 0, SDR, 0)                   # a3
    g.AudioConnection(preProcessor, 1, SDR, 1)                   # a4
    g.AudioConnection(SDR, 0, Q_in_L, 0)                         # c5
    g.AudioConnection(SDR, 1, Q_in_R, 0)                         # c6
    g.AudioConnection(Q_out_L, 0, AudioFFT, 0)                   # c6a
    g.AudioConnection(Q_out_L, 0, audio_out, 0)                  # c7
    g.AudioConnection(Q_out_R, 0, audio_out, 1)                  # c8
    Q_in_L.begin(); Q_in_R.begin()                               # CONV:205-206
    rf, af = 0, 0
    for b in range(nblk + 10):
        if b < nblk:
            blk = iq[0, b * 128:(b + 1) * 128]
            IQinput.push(blk[None, :, 0], blk[None, :, 1])
        assert g.update_all() == 0
        # loop(): the convolutional stage lives in the engine node here, so the queues just hand over
        while Q_in_L.available() > 0 and Q_in_R.available() > 0:
            l, r = Q_in_L.readBuffer().copy(), Q_in_R.readBuffer().copy()
            Q_in_L.freeBuffer(); Q_in_R.freeBuffer()
            ol, orr = Q_out_L.getBuffer(), Q_out_R.getBuffer()
            ol[:] = l; orr[:] = r
            assert Q_out_L.playBuffer() == 0 and Q_out_R.playBuffer() == 0
            assert g.update_all() == 0                           # the ISR tick that plays the pair
        rf += FFT.available()
        af += AudioFFT.available()
    for node in (SDR, FFT, AudioFFT, biquad1, biquad2):
        assert node.status() == 0
    got = np.concatenate(heard)
    r16, _ = oracle.OracleChain(**K1).process(iq[0])
    assert got.shape == r16.shape and np.abs(got.astype(np.int32) - r16.astype(np.int32)).max() <= 1
    assert rf >= 3 and af >= 3                                   # 96 IQ ticks / 30; 24 audio blocks: frames at 8, 12, ...
    spec = AudioFFT.output()[0].astype(int)
    # what the display does with the two analysers: read(bin) and read(first, last) (FFTIQ.h:70-86 keeps binLast out,
    # the Teensy library's 1024-point analyser includes it)
    pan = FFT.output()[0].astype(int)
    assert FFT.read(0, 80) == pan[80] / 16384.0 and FFT.read(0, 75, 85) == float(np.float32(pan[75:85].sum())) / 16384.0
    assert AudioFFT.read(0, 30) == spec[30] / 16384.0 and AudioFFT.read(0, 28, 32) == float(np.float32(spec[28:33].sum())) / 16384.0
    assert FFT.read(0, 256) == 0.0 and AudioFFT.read(0, 512) == 0.0 and FFT.read(1, 3) == 0.0     # no such bin / channel
    for f in (700.0, 1000.0, 1900.0):                            # the USB tones of the synthetic input, 24 kHz / 1024 per bin
        k = int(round(f / (24000.0 / 1024)))
        assert spec[k - 1:k + 2].max() > 8 * np.median(spec[:128])
    assert g.memory_usage()[0] == 0 and g.memory_usage()[1] <= 40


def test_queue_rings_have_the_sizes_of_the_references_image(rdsp):
    """AudioRecordQueue is a ring of 209, AudioPlayQueue one of 80 in the Teensy 4 build the reference ships (its
    available() computes head + 209 - tail, its playBuffer() wraps past 79): a record queue nobody reads keeps 208
    blocks and drops the rest, a play queue takes 79 before it reports that the library would spin."""
    from radiodsp_sdr_rx_amd.graph import Graph
    g = Graph(1)
    g.AudioMemory(400)
    src = g.input_node()
    q = g.record_queue()
    g.AudioConnection(src, 0, q, 0)
    q.begin()
    z = np.zeros((1, 128), np.int16)
    for t in range(230):
        src.push(z, z)
        assert g.update_all() == 0
    assert q.available() == 208
    assert g.memory_usage()[0] <= 209 + 2
    p = g.play_queue()
    n = 0
    while n < 100:
        buf = p.getBuffer()
        assert buf is not None
        if p.playBuffer() != 0:
            break
        n += 1
    assert n == 79
