"""Soak run of the randomised control sessions (tests/test_random_sessions.py) over many seeds:
python tests/micro/session_soak.py <first_seed> <count> [channels].  Not part of the suite.
The bitwise kinds (plain / pipelined / re-partitioned, checkpoint / resume) must never fail.  The two
that compare float32 implementations have statistical tails: in 400 sessions each, one oracle session had a
spectral-threshold hop of 1.9e-3, and two switching sessions missed the truth-anchored criterion by a hair
(worst channel 1.44e-5 against 1.5 x 9.46e-6; one channel at 3.9 x its own oracle distance)."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import radiodsp_sdr_rx_amd as R
import importlib.util
spec = importlib.util.spec_from_file_location("trs", "/root/repo/tests/test_random_sessions.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from cases import K3
from radiodsp_sdr_rx_amd.chain import Chain, synth_iq
import os
if os.environ.get("RDSP_SOAK_FIR"):   # 0 / 2: every chain of the soak runs the direct form / the 448-sample frames (default: the library's granule frames)
    Chain.default_fir_variant = int(os.environ["RDSP_SOAK_FIR"])
first, count = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:          # more channels: longer kernels, the host runs further ahead of the device
    m.NCH = int(sys.argv[3])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    four = seed % 5 == 0
    cfg = dict(K3, fft_l=2048) if four else K3
    ops = m.make_script(rng, n_ops=30 if four else 40, granule=32 if four else 8)
    total = sum(op[1] for op in ops if op[0] == "proc")
    iq = synth_iq(m.NCH, total * 128)
    for c in range(0, m.NCH, 7):
        for p in rng.integers(2000, iq.shape[1] - 4, 12):
            iq[c, p:p + 3] = 30000
    plain = m.run_script(R, torch, ops, iq, m.NCH, False, 0, cfg)
    piped = m.run_script(R, torch, ops, iq, m.NCH, True, 64, cfg)
    part = m.run_script(R, torch, ops, iq, 63, True, 0, cfg)
    if m.NCH > 150:            # and once more with everything on the caller's stream but the device kept busy
        plain2 = m.run_script(R, torch, ops, iq, m.NCH, False, 0, cfg)
        if not all(np.array_equal(a, b, equal_nan=True) for a, b in zip(plain, plain2)):
            print("PLAIN RUNS DIFFER seed", seed, flush=True)
    ok = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(plain, piped)) and \
         all(np.array_equal(a[:63], b, equal_nan=True) for a, b in zip(plain, part))
    if not ok:
        bad += 1
        names = ("audio", "scalars", "nr_w", "als_w")
        what = [n + ":piped" for n, a, b in zip(names, plain, piped) if not np.array_equal(a, b, equal_nan=True)] + \
               [n + ":part" for n, a, b in zip(names, plain, part) if not np.array_equal(a[:63], b, equal_nan=True)]
        det = ""
        for tag, other, sl in (("piped", piped, slice(None)), ("part", part, slice(0, 63))):
            a, b = plain[0][sl], other[0]
            d = (a != b).any(axis=2)
            if d.any():
                chs = np.where(d.any(axis=1))[0]
                det += f" [{tag}: {len(chs)} channels e.g. {chs[:6]}, first sample {[int(np.argmax(d[c])) for c in chs[:4]]}, max abs {np.abs(a.astype(int) - b.astype(int)).max()}]"
        print("MISMATCH seed", seed, what, det, "nan in plain scalars/w:", bool(np.isnan(plain[1]).any()), bool(np.isnan(plain[3]).any()), flush=True)
    if (seed - first) % 10 == 9:
        print("seeds", first, "..", seed, "done,", bad, "mismatches", flush=True)
print("soak:", count, "sessions,", bad, "mismatches")

if len(sys.argv) > 4 and sys.argv[4] == "bitwise":   # only the kinds that must never fail
    for seed in range(first, first + count):
        try:
            m.test_random_session_resumed_from_a_checkpoint_is_bit_exact(R, seed)
        except AssertionError as e:
            print("FAIL resume seed", seed, flush=True)
    print("soak resume:", count, "sessions done")
    sys.exit(0)
# the oracle-checked sessions and the truth-anchored switching sessions over the same seed range
import oracle_lib
oracle_lib.build()
fails = {"oracle": [], "switching": [], "resume": []}
for seed in range(first, first + count):
    for name, fn in (("oracle", m.test_random_retune_session_matches_oracle),
                     ("switching", m.test_random_nr_and_notch_switching_is_truth_anchored),
                     ("resume", m.test_random_session_resumed_from_a_checkpoint_is_bit_exact)):
        try:
            fn(R, seed) if name == "resume" else fn(R, oracle_lib, seed)
        except AssertionError as e:
            fails[name].append(seed)
            print("FAIL", name, "seed", seed, str(e)[:300].replace("\n", " "), flush=True)
print("soak oracle / switching / resume:", count, "sessions each, failures", fails)
