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


def main():
    oracle_lib.build()
    for name, case in GOLDEN_CASES.items():
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
        if cfg.get("lms_nr", 0) > 0 or cfg.get("als_mode", "off") != "off":
            # NLMS chains: the float64 evaluation of the same chain (tests/np_model.py) travels with
            # the fixture; the GPU test anchors its tolerance on it (truth), not on the float32 oracle
            extra["out_f64"] = np.stack([np_model.Model(**cfg).process(iq[c]) for c in range(case["channels"])])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), iq=iq, out_i16=np.stack(o16),
                            out_f32=np.stack(o32), **extra)
        print(name, iq.shape, np.stack(o16).shape)


if __name__ == "__main__":
    main()
