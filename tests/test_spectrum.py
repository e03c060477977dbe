"""IQ panadapter spectrum analyser (AudioAnalyzeFFT256IQ).
CPU: known-answer tests of the oracle's integer restatement.  GPU: bit-exact parity of
the batched kernel through the C-ABI (integer path => exact, split calls included)."""
import ctypes as C

import numpy as np
import pytest

I16P = C.POINTER(C.c_int16)


def _olib(oracle):
    lib = oracle.load()
    lib.orc_fft256iq_create.restype = C.c_void_p
    lib.orc_fft256iq_create.argtypes = [C.c_int, C.c_int]
    lib.orc_fft256iq_destroy.argtypes = [C.c_void_p]
    lib.orc_fft256iq_update.argtypes = [C.c_void_p, I16P, I16P]
    lib.orc_fft256iq_update.restype = C.c_int
    lib.orc_fft256iq_output.restype = C.POINTER(C.c_uint16)
    lib.orc_fft256iq_output.argtypes = [C.c_void_p]
    lib.orc_fft256iq_averageTogether.argtypes = [C.c_void_p, C.c_int]
    lib.orc_fft256iq_windowFunction.argtypes = [C.c_void_p, C.c_int]
    lib.orc_fft256iq_windowFunction_table.argtypes = [C.c_void_p, I16P]
    lib.orc_fft256iq_read.restype = C.c_float
    lib.orc_fft256iq_read.argtypes = [C.c_void_p, C.c_uint]
    lib.orc_fft256iq_read_range.restype = C.c_float
    lib.orc_fft256iq_read_range.argtypes = [C.c_void_p, C.c_uint, C.c_uint]
    lib.orc_cfft_radix4_q15_256.argtypes = [I16P]
    lib.orc_window_q15.argtypes = [C.c_int, I16P]
    lib.orc_sqrt_uint32.argtypes = [C.c_uint32]
    lib.orc_sqrt_uint32.restype = C.c_uint32
    return lib


def oracle_spectra(lib, iq, naverage, window):
    """iq int16 [n, 2] (n multiple of 128) -> list of uint16[256] spectra, in order.  window: an id, or an int16
    table handed over the way the reference does (windowFunction(const int16_t *), FFTIQ.h:93)."""
    if isinstance(window, np.ndarray):
        s = lib.orc_fft256iq_create(naverage, 0)
        lib.orc_fft256iq_windowFunction_table(s, np.ascontiguousarray(window, np.int16).ctypes.data_as(I16P))
    else:
        s = lib.orc_fft256iq_create(naverage, window)
    outs = []
    i = np.ascontiguousarray(iq[:, 0])
    q = np.ascontiguousarray(iq[:, 1])
    for b in range(len(iq) // 128):
        if lib.orc_fft256iq_update(s, i[b * 128:].ctypes.data_as(I16P), q[b * 128:].ctypes.data_as(I16P)):
            outs.append(np.ctypeslib.as_array(lib.orc_fft256iq_output(s), (256,)).copy())
    lib.orc_fft256iq_destroy(s)
    return outs


def test_fixed_point_fft_tracks_float_dft(oracle):
    lib = _olib(oracle)
    rng = np.random.default_rng(2)
    x = rng.integers(-20000, 20000, size=(256, 2)).astype(np.int16)
    buf = x.reshape(-1).copy()
    lib.orc_cfft_radix4_q15_256(buf.ctypes.data_as(I16P))
    X = np.fft.fft(x[:, 0].astype(float) + 1j * x[:, 1]) / 256
    got = buf[0::2] + 1j * buf[1::2]
    assert np.abs(got - X).max() < 8  # scaled fixed-point: a few LSB of truncation noise
    assert lib.orc_sqrt_uint32(0) == 0 and lib.orc_sqrt_uint32(15) == 3 and lib.orc_sqrt_uint32(16) == 4
    assert lib.orc_sqrt_uint32(4294967295) == 65535
    w = np.zeros(256, np.int16)
    lib.orc_window_q15(1, w.ctypes.data_as(I16P))
    assert w[0] == 0 and w[127] == w[128] == 32767 and np.array_equal(w, w[::-1])   # symmetric about 127.5: i / (N - 1)
    # the published routine in a second, independent form: plain-integer radix-4 DIF with the stage scalings of
    # arm_radix4_butterfly_q15 written out per component
    assert np.array_equal(buf, _cmsis_radix4_q15_model(x.reshape(-1), 256))
    rails = rng.choice(np.array([-32768, 32767], np.int16), size=2048)
    b2 = rails.copy()
    lib.orc_cfft_radix4_q15_n.argtypes = [I16P, C.c_int]
    lib.orc_cfft_radix4_q15_n(b2.ctypes.data_as(I16P), 1024)
    assert np.array_equal(b2, _cmsis_radix4_q15_model(rails, 1024))                # saturating paths included


def _cmsis_radix4_q15_model(buf, n):
    """Python integers; natural-order radix-4 decimation in frequency, outputs k = 0..3 of a butterfly at
    base + k L, digit reversal at the end -- the data flow of the GPU kernels, not CMSIS' in-place order."""
    import os
    tw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "firmware_tables.npz"))["twiddle_q15_4096"]
    sat = lambda v: max(-32768, min(32767, v))
    re = [int(v) for v in buf[0::2]]
    im = [int(v) for v in buf[1::2]]
    stages = {256: 4, 1024: 5}[n]
    L = n // 4
    for st in range(stages):
        first, last = st == 0, st == stages - 1
        for g in range(0, n, 4 * L):
            for j in range(L):
                idx = [g + j + k * L for k in range(4)]
                a, b, c, d = [(re[i] >> 2, im[i] >> 2) if first else (re[i], im[i]) for i in idx]
                R = (sat(a[0] + c[0]), sat(a[1] + c[1]))
                S = (sat(a[0] - c[0]), sat(a[1] - c[1]))
                V = (sat(b[0] + d[0]), sat(b[1] + d[1]))
                T = (sat(b[0] - d[0]), sat(b[1] - d[1]))
                if first:
                    y0 = ((R[0] + V[0]) >> 1, (R[1] + V[1]) >> 1)
                    y2 = (sat(R[0] - V[0]), sat(R[1] - V[1]))
                    y1 = (sat(S[0] + T[1]), sat(S[1] - T[0]))
                    y3 = (sat(S[0] - T[1]), sat(S[1] + T[0]))
                elif not last:
                    y0 = (((R[0] + V[0]) >> 1) >> 1, ((R[1] + V[1]) >> 1) >> 1)
                    y2 = ((R[0] - V[0]) >> 1, (R[1] - V[1]) >> 1)
                    y1 = ((S[0] + T[1]) >> 1, (S[1] - T[0]) >> 1)
                    y3 = ((S[0] - T[1]) >> 1, (S[1] + T[0]) >> 1)
                else:
                    y0 = ((R[0] + V[0]) >> 1, (R[1] + V[1]) >> 1)
                    y2 = ((R[0] - V[0]) >> 1, (R[1] - V[1]) >> 1)
                    y1 = ((S[0] + T[1]) >> 1, (S[1] - T[0]) >> 1)
                    y3 = ((S[0] - T[1]) >> 1, (S[1] + T[0]) >> 1)
                ys = [y0, y1, y2, y3]
                if not last:
                    for k in (1, 2, 3):
                        m = k * j * (n // (4 * L)) * (4096 // n)
                        co, si = int(tw[2 * m]), int(tw[2 * m + 1])
                        yr, yi = ys[k]
                        ys[k] = ((co * yr + si * yi) >> 16, (co * yi - si * yr) >> 16)
                for k in range(4):
                    re[idx[k]], im[idx[k]] = ys[k]
        L //= 4
    out = np.zeros(2 * n, np.int16)
    for p in range(n):
        k, q = 0, p
        for _ in range(stages):
            k = (k << 2) | (q & 3)
            q >>= 2
        out[2 * k], out[2 * k + 1] = re[p], im[p]
    return out


def test_tone_lands_on_the_reference_bin_order(oracle):
    """output[255 - (i ^ 128)] (FFTIQ.cpp:105): bin +20 of the IQ stream is index 107;
    naverage frames are averaged (sum of |X|^2 / n, then sqrt), first block only primes."""
    lib = _olib(oracle)
    n = np.arange(128 * 9)
    x = 8000 * np.exp(2j * np.pi * (20 / 256) * n)
    iq = np.stack([np.round(x.real), np.round(x.imag)], 1).astype(np.int16)
    outs = oracle_spectra(lib, iq, 4, 0)
    assert len(outs) == 2  # 9 blocks -> 8 frames -> 2 outputs
    assert outs[0].argmax() == 255 - (20 ^ 128) == 107
    # sum_n (|X|^2 / n) with |X| ~ 8000 - a few LSB; sqrt back to ~ |X|
    assert abs(int(outs[0][107]) - 8000) < 16
    assert np.sort(outs[0])[-2] < 40


@pytest.mark.gpu
@pytest.mark.parametrize("naverage,window,calls", [(1, "none", 1), (8, "AudioWindowHanning256", 1),
                                                    (30, "AudioWindowHanning256", 3), (5, "AudioWindowBlackmanHarris256", 4),
                                                    (8, "AudioWindowBlackmanNuttall256", 2), (3, "AudioWindowFlattop256", 1)])
def test_gpu_spectrum_is_bit_exact(rdsp, oracle, naverage, window, calls):
    import torch
    assert torch.cuda.is_available()
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.spectrum import WINDOWS, AnalyzeFFT256IQ
    lib = _olib(oracle)
    nch, nblk = 10, 96
    iq = synth_iq(nch, nblk * 128)
    iq[3] = np.clip(iq[3].astype(np.int32) * 4, -32768, 32767).astype(np.int16)  # drive saturation paths
    iq[4, :, :] = -32768
    # rail-to-rail data: butterfly outputs of +-32768 on both components meet 45-degree twiddles,
    # the only way a product leaves the int16 range; and the whole int16 range uniformly
    rng = np.random.default_rng(12)
    iq[7] = rng.choice(np.array([-32768, 32767], np.int16), size=iq[7].shape)
    iq[8] = rng.integers(-32768, 32768, size=iq[8].shape).astype(np.int16)
    iq[9] = np.repeat(rng.choice(np.array([-32768, 32767], np.int16), size=(nblk * 128 // 64, 2)), 64, axis=0)
    fft = AnalyzeFFT256IQ(nch, naverage=naverage, window=window)
    got = []
    step = nblk // calls * 128
    for k in range(calls):
        part = torch.from_numpy(np.ascontiguousarray(iq[:, k * step:(k + 1) * step])).cuda()
        o = fft.update(part)
        torch.cuda.synchronize()
        got.append(o.cpu().numpy().view(np.uint16))
    got = np.concatenate(got, axis=1)
    for c in range(nch):
        ref = oracle_spectra(lib, iq[c], naverage, WINDOWS[window])
        assert got.shape[1] == len(ref)
        assert np.array_equal(got[c], np.stack(ref)) if ref else got.shape[1] == 0
    if got.shape[1]:
        assert fft.available() and not fft.available()
        assert fft.read(0, 300) == 0.0
        assert fft.read(0, 5) == got[0, -1, 5] / 16384.0


@pytest.mark.gpu
def test_gpu_spectrum_with_the_firmware_tables_like_the_sketch(rdsp, oracle):
    """`FFT.windowFunction(AudioWindowHanning256); FFT.averageTogether(30);` (INO:144-145) with the table the
    reference's firmware image holds handed over by pointer (FFTIQ.h:93-95), the constructor's own settings
    (BlackmanNuttall256, naverage 8, FFTIQ.h:55-58) and windowFunction(NULL); read(bin), read(first, last)."""
    import os
    import torch
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.spectrum import WINDOWS, AnalyzeFFT256IQ, window_q15
    lib = _olib(oracle)
    fw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "firmware_tables.npz"))
    nch, nblk = 4, 61
    iq = synth_iq(nch, nblk * 128)
    iq[3] = np.random.default_rng(5).integers(-32768, 32768, size=iq[3].shape).astype(np.int16)
    dev = torch.from_numpy(iq).cuda()
    # the constructor's defaults, untouched
    fft = AnalyzeFFT256IQ(nch)
    got = fft.update(dev).cpu().numpy().view(np.uint16)
    for c in range(nch):
        assert np.array_equal(got[c], np.stack(oracle_spectra(lib, iq[c], 8, fw["blackman_nuttall256"])))
        assert np.array_equal(got[c], np.stack(oracle_spectra(lib, iq[c], 8, WINDOWS["AudioWindowBlackmanNuttall256"])))
    # the sketch's setup() calls, table by pointer
    fft = AnalyzeFFT256IQ(nch)
    fft.windowFunction(fw["hann256"])
    fft.averageTogether(30)
    got = fft.update(dev).cpu().numpy().view(np.uint16)
    assert got.shape[1] == 2
    for c in range(nch):
        assert np.array_equal(got[c], np.stack(oracle_spectra(lib, iq[c], 30, fw["hann256"])))
    assert np.array_equal(window_q15(1), fw["hann256"])
    row = got[0, -1]
    assert fft.read(0, 12) == row[12] / 16384.0
    assert fft.read(0, 75, 85) == float(np.float32(int(row[75:85].sum()))) / 16384.0     # `while (binFirst < binLast)`: bin 85 is not added
    assert fft.read(0, 9, 9) == row[9] / 16384.0 and fft.read(0, 300, 400) == 0.0
    # an arbitrary caller table, then no window at all
    tab = np.random.default_rng(6).integers(-32768, 32768, 256).astype(np.int16)
    fft = AnalyzeFFT256IQ(nch, naverage=2)
    fft.windowFunction(tab)
    got = fft.update(dev[:, :33 * 128].contiguous()).cpu().numpy().view(np.uint16)
    fft.windowFunction(None)
    got2 = fft.update(dev[:, 33 * 128:].contiguous()).cpu().numpy().view(np.uint16)
    for c in range(nch):
        o = lib.orc_fft256iq_create(2, 0)
        lib.orc_fft256iq_windowFunction_table(o, tab.ctypes.data_as(I16P))
        want = []
        for b in range(nblk):
            if b == 33:
                lib.orc_fft256iq_windowFunction_table(o, None)
            i, q = np.ascontiguousarray(iq[c, b * 128:(b + 1) * 128, 0]), np.ascontiguousarray(iq[c, b * 128:(b + 1) * 128, 1])
            if lib.orc_fft256iq_update(o, i.ctypes.data_as(I16P), q.ctypes.data_as(I16P)):
                want.append(np.ctypeslib.as_array(lib.orc_fft256iq_output(o), (256,)).copy())
        lib.orc_fft256iq_destroy(o)
        assert np.array_equal(np.concatenate([got[c], got2[c]]), np.stack(want))


@pytest.mark.gpu
def test_analyser_as_a_graph_node_like_the_sketch(rdsp, oracle):
    """IQinput -> (I, Q) -> FFT node, ticked block by block (INO:52,57,71-74): every
    completed average equals the oracle's, available() fires once per output."""
    import torch
    assert torch.cuda.is_available()
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.graph import Graph
    from radiodsp_sdr_rx_amd.spectrum import WINDOWS, AnalyzeFFT256IQ
    lib = _olib(oracle)
    nch, nblk, nav = 3, 25, 4
    iq = synth_iq(nch, nblk * 128)
    g = Graph(nch)
    g.AudioMemory(12)
    src = g.input_node()
    fft = g.spectrum_node(AnalyzeFFT256IQ(nch, naverage=nav, window="AudioWindowHanning256"))
    g.AudioConnection(src, 0, fft, 0)
    g.AudioConnection(src, 1, fft, 1)
    outs = []
    for b in range(nblk):
        blk = iq[:, b * 128:(b + 1) * 128]
        src.push(blk[:, :, 0], blk[:, :, 1])
        g.update_all()
        assert fft.status() == 0
        if fft.available():
            outs.append(fft.output())
            assert not fft.available()
    assert g.memory_usage()[0] == 0          # (blocks in use, peak)
    for c in range(nch):
        ref = oracle_spectra(lib, iq[c], nav, WINDOWS["AudioWindowHanning256"])
        assert len(outs) == len(ref) == (nblk - 1) // nav
        for k, r in enumerate(ref):
            assert np.array_equal(outs[k][c], r)
    assert fft.read(0, 10) == outs[-1][0, 10] / 16384.0
    assert fft.read(2, 10, 14) == float(np.float32(int(outs[-1][2, 10:14].sum()))) / 16384.0


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_gpu_spectrum_random_sessions_are_bit_exact(rdsp, oracle, seed):
    """update() calls of random length with averageTogether() / windowFunction() in between, against the
    restatement block by block.  averageTogether below the running frame count lets the reference's
    `uint8_t count` (FFTIQ.h:105) run on to its 8-bit wrap, with a silent restart of the sums there
    (FFTIQ.cpp:88-93,99-100): script 1 forces that case; available()/outputs follow."""
    import torch
    assert torch.cuda.is_available()
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.spectrum import WINDOWS, AnalyzeFFT256IQ
    lib = _olib(oracle)
    rng = np.random.default_rng(seed)
    names = list(WINDOWS)
    nch = 3
    ops = []
    for _ in range(14):
        kind = str(rng.choice(["update", "update", "avg", "win"]))
        if kind == "update":
            ops.append(("update", int(rng.integers(1, 40))))
        elif kind == "avg":
            ops.append(("avg", int(rng.choice([1, 2, 3, 5, 8, 30, 200]))))
        else:
            ops.append(("win", str(rng.choice(names))))
    if seed == 1:   # 30 frames into an average of 200, then averageTogether(5): count = 30 > 5 runs to the wrap
        ops = [("avg", 200), ("update", 31), ("avg", 5), ("update", 100), ("update", 140), ("update", 20)] + ops
    ops.append(("update", 64))
    total = sum(op[1] for op in ops if op[0] == "update")
    iq = synth_iq(nch, total * 128)
    iq[1] = rng.integers(-32768, 32768, size=iq[1].shape).astype(np.int16)
    fft = AnalyzeFFT256IQ(nch, naverage=8, window=names[0])
    ors = [lib.orc_fft256iq_create(8, WINDOWS[names[0]]) for _ in range(nch)]
    dev = torch.from_numpy(iq).cuda()
    pos = 0
    for op in ops:
        if op[0] == "update":
            n = op[1]
            got = fft.update(dev[:, pos * 128:(pos + n) * 128].contiguous()).cpu().numpy().view(np.uint16)
            for c in range(nch):
                want = []
                i = np.ascontiguousarray(iq[c, pos * 128:(pos + n) * 128, 0])
                q = np.ascontiguousarray(iq[c, pos * 128:(pos + n) * 128, 1])
                for b in range(n):
                    if lib.orc_fft256iq_update(ors[c], i[b * 128:].ctypes.data_as(I16P), q[b * 128:].ctypes.data_as(I16P)):
                        want.append(np.ctypeslib.as_array(lib.orc_fft256iq_output(ors[c]), (256,)).copy())
                assert got.shape[1] == len(want), (seed, op, got.shape, len(want))
                if want:
                    assert np.array_equal(got[c], np.stack(want)), (seed, op, c)
            pos += n
        elif op[0] == "avg":
            fft.averageTogether(op[1])
            for o in ors:
                lib.orc_fft256iq_averageTogether(o, op[1])
        else:
            fft.windowFunction(op[1])
            for o in ors:
                lib.orc_fft256iq_windowFunction(o, WINDOWS[op[1]])
    for o in ors:
        lib.orc_fft256iq_destroy(o)
