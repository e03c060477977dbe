"""Generates the committed golden vectors in tests/golden/*.npz.

The reference ships no fixtures and cannot run here, so these are outputs of the
CPU oracle (oracle/rdsp_oracle.c) on deterministic synthetic IQ: they pin the
oracle against accidental change and give the GPU tests fixed expected outputs
that travel to the GPU box.  Fixtures of chains with an NLMS stage also carry the float64
result of tests/np_model.py (out_f64).  Regenerate with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import np_model  # noqa: E402
import oracle_lib  # noqa: E402
from cases import GOLDEN_CASES, add_impulses, apply_setup  # noqa: E402
from radiodsp_sdr_rx_amd.chain import synth_iq  # noqa: E402


def make_nodes():
    """graph nodes beside the chain (integer analysers, int16 biquad): nodes.npz holds one input and what
    the oracle's restatements give for it -- the panadapter analyser (naverage 5, AudioWindowHanning256), the
    audio analyser on the I side (AudioWindowHanning1024; both tables as in the reference's firmware image) and
    `setHighpass(0, 500, 0.5)` + a notch in stage 1 on the I side (the Teensy library's fixed-point AudioFilterBiquad)"""
    import ctypes as C
    lib = oracle_lib.load()
    I16P, F32P = C.POINTER(C.c_int16), C.POINTER(C.c_float)
    nch, nblk = 2, 40
    iq = synth_iq(nch, nblk * 128)
    iq[1] = np.clip(iq[1].astype(np.int32) * 4, -32768, 32767).astype(np.int16)
    lib.orc_fft256iq_create.restype = C.c_void_p
    lib.orc_fft256iq_create.argtypes = [C.c_int, C.c_int]
    lib.orc_fft256iq_update.argtypes = [C.c_void_p, I16P, I16P]
    lib.orc_fft256iq_output.restype = C.POINTER(C.c_uint16)
    lib.orc_fft256iq_output.argtypes = [C.c_void_p]
    lib.orc_fft256iq_destroy.argtypes = [C.c_void_p]
    lib.orc_fft1024_create.restype = C.c_void_p
    lib.orc_fft1024_create.argtypes = [C.c_int]
    lib.orc_fft1024_update.argtypes = [C.c_void_p, I16P]
    lib.orc_fft1024_output.restype = C.POINTER(C.c_uint16)
    lib.orc_fft1024_output.argtypes = [C.c_void_p]
    lib.orc_fft1024_destroy.argtypes = [C.c_void_p]
    lib.orc_biquad_design.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, F32P]
    lib.orc_float_to_q15.argtypes = [F32P, I16P, C.c_uint32]

    class OrcBiquad(C.Structure):
        _fields_ = [("n_stages", C.c_int), ("coef", C.c_float * 20), ("state", C.c_float * 16)]
    lib.orc_biquad_init.argtypes = [C.POINTER(OrcBiquad), C.c_int, F32P]
    lib.orc_biquad_run.argtypes = [C.POINTER(OrcBiquad), F32P, C.c_int]
    lib.orc_biquad_set_stage.argtypes = [C.POINTER(OrcBiquad), C.c_int, F32P]
    spec256, spec1024, bq = [], [], []
    for c in range(nch):
        i = np.ascontiguousarray(iq[c, :, 0])
        q = np.ascontiguousarray(iq[c, :, 1])
        s = lib.orc_fft256iq_create(5, 1)
        outs = []
        for b in range(nblk):
            if lib.orc_fft256iq_update(s, i[b * 128:].ctypes.data_as(I16P), q[b * 128:].ctypes.data_as(I16P)):
                outs.append(np.ctypeslib.as_array(lib.orc_fft256iq_output(s), (256,)).copy())
        lib.orc_fft256iq_destroy(s)
        spec256.append(np.stack(outs))
        s = lib.orc_fft1024_create(1)
        outs = []
        for b in range(nblk):
            if lib.orc_fft1024_update(s, i[b * 128:].ctypes.data_as(I16P)):
                outs.append(np.ctypeslib.as_array(lib.orc_fft1024_output(s), (512,)).copy())
        lib.orc_fft1024_destroy(s)
        spec1024.append(np.stack(outs))
        from test_audio_nodes import TeensyBiquadOracle
        o = TeensyBiquadOracle(lib)
        o.set(0, "highpass", 500.0, 0.5)
        o.set(1, "notch", 1000.0, 4.0)
        r16 = o.update(i)
        bq.append(r16)
    np.savez_compressed(os.path.join(HERE, "nodes.npz"), iq=iq, spectrum256=np.stack(spec256),
                        fft1024=np.stack(spec1024), biquad=np.stack(bq))
    print("nodes", iq.shape, np.stack(spec256).shape, np.stack(spec1024).shape)


def main():
    oracle_lib.build()
    only = set(sys.argv[1:])   # e.g. `make_golden.py nodes` regenerates that fixture alone
    if not only or "nodes" in only:
        make_nodes()
    for name, case in GOLDEN_CASES.items():
        if only and name not in only:
            continue
        iq = synth_iq(case["channels"], case["blocks"] * 128, cw=case.get("cw", False))
        if case.get("impulses"):
            iq = add_impulses(iq)
        o16, o32 = [], []
        for c in range(case["channels"]):
            ch = oracle_lib.OracleChain(**case["cfg"])
            apply_setup(ch, case.get("setup"), oracle=True)
            a, b = ch.process(iq[c])
            o16.append(a)
            o32.append(b)
        extra = {}
        cfg = case["cfg"]
        if cfg.get("lms_nr", 0) > 0 or cfg.get("als_mode", "off") != "off" or cfg.get("demod") == "SAM":
            # recursive stages (NLMS, SAM PLL): the float64 evaluation of the same chain (tests/np_model.py) travels with
            # the fixture; the GPU test anchors its tolerance on it (truth), not on the float32 oracle
            extra["out_f64"] = np.stack([np_model.Model(**cfg).process(iq[c]) for c in range(case["channels"])])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), iq=iq, out_i16=np.stack(o16),
                            out_f32=np.stack(o32), **extra)
        print(name, iq.shape, np.stack(o16).shape)


if __name__ == "__main__":
    main()
