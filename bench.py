#!/usr/bin/env python3
"""bench.py -- throughput of the per-block IQ receive chain on MI355X.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (launched by torch.distributed.run for N > 1).  Channels are
sharded by rank with no data-path collective (RCCL unused; the only
communication is the timing barrier / max-reduce, over gloo).  A step = one pass
of the hot path over one batch: `--blocks` input blocks of 128 IQ samples for
every channel of the rank, already resident in HBM.  Rank 0 prints ONE JSON
line.  Default workload = BASELINE.json configs[2] (K3, "full SSB+NR chain":
4096 channels, USB + 512-pt spectral NR + LMS auto-notch + AGC), the
configuration the metric is quoted on; --config K2|K4|K5 selects the others.
The default run (no --config) times the K3 headline leg and then, each the same
way, K2, the K5 per-GPU shape and K4 (`configs` in the line; N > 1: K5 only).
Stage A3 runs in the throughput form (rdsp_chain_set_fir_variant 2, named in
`config.decimator`); every leg is preceded by un-timed settle steps
(`setup_steps`) so that the timed steps are a stream in flight.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# flops per input IQ sample of the chain AS RUN (SURVEY Appendix D conventions: FMA = 2, complex FFT
# 5 N log2 N): the decimator in the frequency domain -- four 512-point transforms, 4 x 512 complex
# multiply-accumulates and one inverse per 1792 input samples, 73 flop per sample where the direct
# 256-tap form is 256 -- mixer ~14, overlap-save FFT pair + mask 42-62, spectral NR ~8, NLMS 96, AGC/pack ~2
FLOP_FD = (5 * 5 * 512 * 9 + 8 * 4 * 512) / 1792.0
FLOP_PER_SAMPLE = {"K2": FLOP_FD + 14 + 42 + 2, "K3": FLOP_FD + 14 + 47 + 8 + 96 + 2, "K4": FLOP_FD + 14 + 62 + 2,
                   "K5": FLOP_FD + 14 + 47 + 8 + 96 + 2}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="K3", choices=["K1", "K2", "K3", "K4", "K5", "F1", "ENGINE"])
    ap.add_argument("--channels-per-gpu", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=512, help="128-sample input blocks per channel per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-kernel HIP events")
    ap.add_argument("--no-iso", action="store_true",
                    help="skip the extra un-timed run of the kernels without overlap (profiling runs: keeps the "
                         "profiler's per-kernel average to the launches of the timed configuration)")
    ap.add_argument("--groups", type=int, default=1,
                    help="receiver groups (SURVEY F2): channel c in group c %% G, every group with its own pass band")
    ap.add_argument("--retune-every", type=int, default=0,
                    help="with --groups: re-tune one group (PBT step) every N timed steps, without synchronising")
    ap.add_argument("--no-host-io", action="store_true",
                    help="skip the extra PCIe-inclusive leg (host memory -> chain -> host memory)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the K2 / K4 / K5 legs that follow the headline leg of the default run")
    ap.add_argument("--spectral-as-written", action="store_true",
                    help="spectral stage re-synthesis as SPEC:229-232 writes it (atan2 + CMSIS table sine / cosine) instead of "
                         "the exact-arithmetic equivalent X mag'/mag (rdsp_set_spectral_resynthesis)")
    ap.add_argument("--no-pipeline", action="store_true", help="run the tail stage in-stream (no overlap with the next step's front stage)")
    ap.add_argument("--lib", default=None,   # a flag only: no environment variable can put another library under the metric run
                    help="A/B runs: another build of librdsp_hip.so (default: the in-tree one); named in the JSON line")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction / JSON plumbing only, no device work (CPU-side test of --gpus N)")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N
    ranks as a CHILD (`python -m torch.distributed.run`, one process per GPU) and relay rank 0's
    JSON line.  This process has not touched torch or HIP, and it does not exec: a process that
    initialised the GPU must never be replaced on this pool."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores(args.gpus) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout)
        sys.exit(r.returncode or 1)
    print(lines[-1])
    sys.exit(0)


def lib_sha():
    """Hash of the sources librdsp_hip.so is built from: profiles/counters.json carries the hash of the
    library its PMC passes were taken on, and the bench line only quotes those counters for the same
    sources (otherwise roofline.traffic and busy_frac_pmc are null and `counters_note` says why)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "radiodsp_sdr_rx_amd", "csrc")
    files = sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h")) +
                   glob.glob(os.path.join(base, "*.c")) + [os.path.join(base, "Makefile")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_counters(config):
    """(per-kernel counters of `config`, note) from profiles/counters.json, or ({}, reason)"""
    cpath = os.path.join(ROOT, "profiles", "counters.json")
    if not os.path.exists(cpath):
        return {}, "no profiles/counters.json"
    try:
        allc = json.load(open(cpath))
    except Exception as e:
        return {}, f"profiles/counters.json unreadable: {e}"
    ent = allc.get(config, {})
    have, want = ent.get("lib_sha"), lib_sha()
    if have != want:
        return {}, (f"profiles/counters.json[{config}] was taken on library sources {have}, this run is {want}: "
                    "PMC constants not quoted (re-run tests/profile_round.sh + tests/summarize_profiles.py)")
    return {k: v for k, v in ent.items() if isinstance(v, dict)}, None


def algorithmic_bytes_per_sample(decim):
    # int16 I+Q read (4 B) + int16 L+R written per output sample (4/D B)
    return 4.0 + 4.0 / decim


def host_cores(n_gpus):
    """Host threads this job may use: a GPU box gives 16 cores per GPU (the
    container may report the whole host)."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16 * max(1, n_gpus)))


def f1_cpu_baseline(args):
    """cpu_baseline leg of the F1 (panadapter) workload: the oracle's integer analyser."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import numpy as np
    import oracle_lib
    from radiodsp_sdr_rx_amd.chain import synth_iq
    cores = host_cores(1)
    lib = C.CDLL(oracle_lib.build(native=True, out_dir="/tmp"))
    lib.orc_fft256iq_multi.restype = C.c_uint64
    lib.orc_fft256iq_multi.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int16), C.c_int, C.c_int]
    nblk, nch = 512, cores * 16
    iq = synth_iq(nch, nblk * 128)
    best = None
    for _ in range(2):
        t = time.perf_counter()
        lib.orc_fft256iq_multi(30, 1, nch, iq.ctypes.data_as(C.POINTER(C.c_int16)), nblk, cores)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print(json.dumps({"value": nch * nblk * 128 / best / 1e6, "unit": "IQ Msamples/s", "cores": cores, "kind": "port",
                      "sample": f"{nch} channels x {nblk} blocks, AudioAnalyzeFFT256IQ restatement, naverage 30, Hann, "
                                f"gcc -O3 -march=native, OpenMP over channels, best of 2"}))


def f1_main(args):
    """IQ panadapter spectrum analyser batched over channels."""
    import torch
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.spectrum import AnalyzeFFT256IQ
    assert torch.cuda.is_available()
    nch, nblk = args.channels_per_gpu or 4096, args.blocks
    iq = torch.from_numpy(synth_iq(nch, nblk * 128, n_threads=host_cores(1))).cuda()
    fft = AnalyzeFFT256IQ(nch, naverage=30, window="AudioWindowHanning256")  # INO:144-145
    for _ in range(args.warmup):
        fft.update(iq)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        fft.update(iq)
        b.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kms = sum(a.elapsed_time(b) for a, b in ev) / args.steps
    samples = nch * nblk * 128
    value = samples * args.steps / elapsed / 1e6
    achieved = 4.0 * samples / (kms * 1e-3) / 1e9
    traffic, valu = None, None   # PMC passes of tests/profile_round.sh (same workload, same sources), per launch
    ctr, ctr_note = load_counters("F1")
    ent = ctr.get("rdsp_spectrum_kernel")
    if ent and (nch, nblk) == (4096, 512):
        traffic, valu = ent.get("hbm_bytes"), {"insts_per_launch": ent.get("valu_insts"), "busy_frac_pmc": ent.get("valu_busy_frac")}
    res = {"metric": "IQ Msamples/s through the IQ panadapter spectrum analyser (SURVEY 8f row F1)", "value": value,
           "unit": "IQ Msamples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "q15", "data": "synthetic",
           "config": {"workload": f"F1: {nch} channels x {nblk} blocks of 128 int16 IQ samples per step; 256-pt q15 "
                                  "radix-4 FFT per block pair (arm_cfft_radix4_q15 as published, firmware twiddles), AudioWindowHanning256 "
                                  "of the firmware image, 30-frame power average, sqrt_uint32_approx"},
           "roofline": {"bound": "hbm", "kernel": "rdsp_spectrum_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "limiter": "valu",
                        "valu": valu, "counters_note": ctr_note,
                        "note": "algorithmic bytes = 4 B per input sample (the spectra written are < 0.1 %); "
                                "integer VALU issue binds"}}
    if not args.no_cpu_baseline:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--config", "F1"],
                               capture_output=True, text=True, timeout=600)
            res["cpu_baseline"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:
            res["cpu_baseline"] = {"value": None, "unit": "IQ Msamples/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(res))


def k1_main(args):
    """BASELINE.json configs[0] / BASELINE.md K1: 1 channel, 96 kHz IQ, 128-sample blocks, USB, NR / notch
    off -- the CPU reference path (plumbing, no GPU).  The sketch's graph (INO:71-89) on the host block
    graph (rdsp_graph.c): IQinput -> engine node -> record queues -> loop() -> play queues, with the
    oracle's chain (the CPU restatement of the reference) as the engine node's update(); one tick per
    128-sample block like the audio interrupt.  Reported like a cpu_baseline (n_gpus 0): it is the
    reference-shaped path timed, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    import radiodsp_sdr_rx_amd as R
    from radiodsp_sdr_rx_amd.chain import synth_iq
    from radiodsp_sdr_rx_amd.graph import Graph

    cfg = dict(R.K_CONFIGS["K1"]["cfg"])
    lib = oracle_lib.load(oracle_lib.build(native=True, out_dir="/tmp"))
    nblk = max(64, (min(args.blocks, 512) // 4) * 4)
    iq = synth_iq(1, nblk * 128)[0]                      # [n, 2] int16
    chain = oracle_lib.OracleChain(lib=lib, **cfg)
    g = Graph(n_channels=1)
    g.AudioMemory(40)                                    # INO:151
    src = g.input_node()
    held, produced = [], []

    def engine_update(n):                                # preProcessor + SDR of the sketch as one node
        bi, bq = n.receiveReadOnly(0), n.receiveReadOnly(1)
        if bi is None or bq is None:                     # FFTIQ.cpp:72: silent early return
            n.release(bi); n.release(bq)
            return
        held.append(np.stack([bi.data()[0], bq.data()[0]], axis=1))
        n.release(bi); n.release(bq)
        if len(held) == 4:                               # 4 input blocks = one 128-sample audio block at fs/4
            o16, _ = chain.process(np.concatenate(held, axis=0))
            held.clear()
            if len(o16) == 128:
                bl, br = n.allocate(), n.allocate()
                if bl is not None and br is not None:
                    bl.data()[0, :] = o16[:, 0]; br.data()[0, :] = o16[:, 1]
                    n.transmit(bl, 0); n.transmit(br, 1)
                n.release(bl); n.release(br)

    eng = g.node(2, engine_update)
    ql, qr = g.record_queue(), g.record_queue()
    g.AudioConnection(src, 0, eng, 0); g.AudioConnection(src, 1, eng, 1)     # INO:81-82
    g.AudioConnection(eng, 0, ql, 0); g.AudioConnection(eng, 1, qr, 0)       # INO:85-86
    ql.begin(); qr.begin()

    def run_once():
        chain_blocks = 0
        t = time.perf_counter()
        for b in range(nblk):
            blk = iq[b * 128:(b + 1) * 128]
            src.push(np.ascontiguousarray(blk[None, :, 0]), np.ascontiguousarray(blk[None, :, 1]))
            g.update_all()
            while ql.available() > 0 and qr.available() > 0:               # loop(): CONV:231-244
                produced.append(ql.readBuffer()[0].copy()); ql.freeBuffer()
                qr.readBuffer(); qr.freeBuffer()
                chain_blocks += 1
        return time.perf_counter() - t, chain_blocks

    run_once()                                           # warm-up
    best, nout = None, 0
    for _ in range(max(1, min(args.steps, 5))):
        dt, nout = run_once()
        best = dt if best is None else min(best, dt)
    # the oracle alone, same blocks, no graph (what the plumbing costs on top)
    c2 = oracle_lib.OracleChain(lib=lib, **cfg)
    t = time.perf_counter()
    for b in range(0, nblk, 4):
        c2.process(iq[b * 128:(b + 4) * 128])
    alone = time.perf_counter() - t
    print(json.dumps({
        "metric": "IQ Msamples/s through full SSB+NR chain; achieved HBM GB/s vs peak",
        "value": nblk * 128 / best / 1e6, "unit": "IQ Msamples/s", "n_gpus": 0, "steps": nblk, "warmup": nblk,
        "ms_per_step": best / nblk * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"K1: 1 channel x {nblk} blocks of 128 int16 IQ samples @96 kHz, one graph tick per block; "
                               "NCO mix + 256-tap polyphase /4 + 256-pt overlap-save USB filter, NR / notch off",
                   "path": "CPU reference path: oracle chain as the engine node of the host block graph (no GPU)"},
        "roofline": None,
        "cpu_baseline": {"value": nblk * 128 / alone / 1e6, "unit": "IQ Msamples/s", "cores": 1, "kind": "port",
                         "sample": f"the same {nblk} blocks through the oracle chain alone (no graph), 4 blocks per call, "
                                   "gcc -O3 -march=native, one thread"},
        "audio_blocks_out": nout, "realtime_factor": nblk * 128 / best / 96000.0}))


def cpu_baseline_worker(args):
    """Runs in a fresh process (no torch, no second OpenMP runtime): times the
    oracle (-O3 -march=native build made on this host) on a bounded sample of the
    same workload with all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    import radiodsp_sdr_rx_amd as R
    from radiodsp_sdr_rx_amd.chain import synth_iq

    kc = R.K_CONFIGS[args.config]
    cfg = dict(kc["cfg"])
    cores = host_cores(1)
    so = oracle_lib.build(native=True, out_dir="/tmp")
    lib = oracle_lib.load(so)
    # calibrate on a small sample, then size the timed sample to ~cpu-seconds
    nblk = min(args.blocks, 512)
    iq = synth_iq(cores, nblk * 128, cw=kc.get("cw", False))
    t = time.perf_counter()
    oracle_lib.multi_process(iq, n_threads=cores, lib=lib, **cfg)
    dt = time.perf_counter() - t
    rate = iq.shape[0] * iq.shape[1] / dt
    nch = int(max(cores, min(4096, args.cpu_seconds * rate / (nblk * 128))))
    nch = (nch // cores) * cores
    iq = synth_iq(nch, nblk * 128, cw=kc.get("cw", False))
    best, spent, reps = None, 0.0, 0
    while reps < 2 or (spent < args.cpu_seconds and reps < 64):  # ~cpu-seconds of CPU work; host scheduling is noisy: keep the best
        t = time.perf_counter()
        oracle_lib.multi_process(iq, n_threads=cores, lib=lib, **cfg)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
        spent += dt
        reps += 1
    val = nch * nblk * 128 / best / 1e6
    # the same oracle on one thread (case (i)), on a smaller sample
    n1 = max(1, min(64, nch // cores))
    t = time.perf_counter()
    oracle_lib.multi_process(iq[:n1], n_threads=1, lib=lib, **cfg)
    one = n1 * nblk * 128 / (time.perf_counter() - t) / 1e6
    print(json.dumps({"value": val, "unit": "IQ Msamples/s", "cores": cores, "kind": "port",
                      "sample": f"{nch} channels x {nblk} blocks of 128 IQ samples, config {args.config}, "
                                f"oracle gcc -O3 -march=native, OpenMP over channels, best of {reps} passes ({spent:.1f} s of CPU work)",
                      "single_thread_value": one, "single_thread_sample": f"{n1} channels x {nblk} blocks"}))


def gather_per_rank(torch, dist, world, elapsed_s, steps, steady_ms):
    """(max elapsed over ranks, per-rank record).  Every rank's own wall time per step and steady-state period go into
    the line beside the max, so that a scaling curve below N x can be read: one slow rank (a straggler card at its
    own power cap) shows as one large entry, host launch overhead as all of them rising with N."""
    mine = [elapsed_s / steps * 1e3, float(steady_ms) if steady_ms is not None else -1.0]
    if world <= 1:
        return elapsed_s, {"ms_per_step": [mine[0]], "ms_per_step_steady": [steady_ms]}
    t = torch.tensor(mine, dtype=torch.float64)
    parts = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, t)
    per = [float(x[0]) for x in parts]
    steady = [float(x[1]) if float(x[1]) >= 0 else None for x in parts]
    return max(per) * steps / 1e3, {"ms_per_step": per, "ms_per_step_steady": steady}


def dry_run(args, rank, world, dist):
    """No device work: every rank pretends one step takes (1 + rank) ms, then the same barrier /
    max-over-ranks / one-JSON-line protocol as the real run.  Exercised by the CPU-side test of
    `bench.py --gpus 2`; its line is marked so that it can never be read as a measurement."""
    import torch
    kc_channels = args.channels_per_gpu or 4096
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(1e-3 * (1 + rank) * args.steps)
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed, per_rank = gather_per_rank(torch, dist, world, elapsed, args.steps, 1.0 + rank)   # the real run's reduction
    if world > 1:
        dist.destroy_process_group()
    if rank == 0:
        line = {"metric": "dry run of the launcher (no device work)", "value": None, "unit": "IQ Msamples/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "per_rank_ms": per_rank, "data": "dry-run",
                "config": {"workload": f"{args.config}: {kc_channels} channels/GPU (not run)",
                           "sharding": f"channels x{world}, no collectives"}}
        if world > 1 and args.config == "K3":   # the shape of the second leg of a real N > 1 run
            line["k5"] = {"config": "K5", "channels_per_gpu": 8192, "channels_total": 8192 * world, "steps": 0,
                          "ms_per_step": None, "value": None, "unit": "IQ Msamples/s", "note": "dry run"}
        print(json.dumps(line))


DECIMATOR_TEXT = {2: "frequency domain, 448-sample frames (rdsp_chain_set_fir_variant 2)", 0: "direct form (variant 0)",
                  -1: "library default, split-invariant for any call split: frequency domain, one granule per frame beside a "
                      "tail stage, on 16-lane rows (two frames per granule) without one",
                  5: "frequency domain on 16-lane rows, two frames per granule (variant 5, split-invariant)"}
WORKLOAD_TEXT = {"K2": "NCO mix + 256-tap polyphase /4 + 256-pt overlap-save USB filter",
                 "K3": "NCO mix + 256-tap polyphase /4 + 512-pt overlap-save USB filter + spectral NR + LMS auto-notch + AGC",
                 "K4": "NCO mix + 256-tap polyphase /4 + 4096-pt overlap-save CW filter (2049 taps) + AGC",
                 "K5": "K3 chain, 8192 channels/GPU"}


def dominant_kernel(fname, front_avg, tail_avg, decim):
    """(name, avg ms, the kernel's own algorithmic bytes per input sample): the kernel of the step that runs
    LONGER.  Every kernel of the chain is priced with the chain's 5 B per input sample in
    `roofline.achieved`; what the kernel itself moves (the tail kernel: 4 B in + 4 B out per OUTPUT sample)
    goes beside it as `kernel_own`."""
    if tail_avg > front_avg:
        return "rdsp_tail_kernel", tail_avg, 8.0 / decim
    return fname, front_avg, algorithmic_bytes_per_sample(decim)


def extra_leg(name, torch, dist, R, Chain, synth_iq, args, rank, world, local_rank, dev, fir_variant, barrier, iq=None, label=None,
              as_written=False):
    """One more BASELINE.json configuration after the headline leg, timed the same way (inputs resident in HBM,
    warm-up, barrier + synchronize on both sides, max over ranks, per-kernel HIP events): K2 / K4 / the K5
    per-GPU shape, so that the driver's one run carries a number for every GPU configuration."""
    kc = R.K_CONFIGS[name]
    cfg = dict(kc["cfg"])
    nch, nblk, decim = kc["channels"], args.blocks, cfg.get("decim", 4)
    n_samples = nblk * 128
    if iq is None or iq.shape[0] != nch:
        host = synth_iq(nch, n_samples, ch0=rank * nch, cw=kc.get("cw", False), n_threads=host_cores(1))
        iq = torch.from_numpy(host).to(dev)
        del host
    out = torch.empty((nch, n_samples // decim, 2), dtype=torch.int16, device=dev)
    ch = Chain(nch, max_blocks_per_call=nblk, device=local_rank, fir_variant=fir_variant, **cfg)
    ch.set_pipelined(not args.no_pipeline)
    if as_written:
        ch.set_spectral_resynthesis(True)   # SPEC:229-232 as written: atan2, arm_cos_f32 / arm_sin_f32
    steps = max(20, min(args.steps, 60))
    for _ in range(40):   # un-timed: the same settling as the headline leg gets (set-up + warm-up)
        ch.process(iq, out=out)
    ch.flush()
    barrier()
    ch.set_timing(not args.no_kernel_timing)
    t0 = time.perf_counter()
    for _ in range(steps):
        ch.process(iq, out=out)
    ch.flush()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    f_ms, t_ms, calls = ch.get_timing()
    span, _ = ch.get_timing_span()
    ch.set_timing(False)
    fname = ch.front_kernel_name()
    elapsed, per_rank = gather_per_rank(torch, dist, world, elapsed, steps, span / (calls - 1) if calls > 1 else None)
    f_avg, t_avg = f_ms / max(calls, 1), t_ms / max(calls, 1)
    dom, dom_ms, b_own = dominant_kernel(fname, f_avg, t_avg, decim)
    B = algorithmic_bytes_per_sample(decim)
    ach = B * nch * n_samples / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None
    ctr, note = load_counters(name)
    leg = {"config": label or name, "workload": f"{name}: {nch} channels/GPU x {nblk} blocks of 128 int16 IQ samples per step; {WORKLOAD_TEXT[name]}",
           "decimator": DECIMATOR_TEXT.get(fir_variant, DECIMATOR_TEXT[-1]),
           "channels_per_gpu": nch, "channels_total": nch * world, "steps": steps, "ms_per_step": elapsed / steps * 1e3,
           "per_rank_ms": per_rank,
           "ms_per_step_steady": span / (calls - 1) if calls > 1 else None,
           "value": float(world) * nch * n_samples * steps / elapsed / 1e6, "unit": "IQ Msamples/s",
           "kernels_ms_per_step": {fname: f_avg, "rdsp_tail_kernel": t_avg} if t_avg > 0 else {fname: f_avg},
           "roofline": {"bound": "hbm", "limiter": "valu", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": ctr.get(dom, {}).get("hbm_bytes"),
                        "counters_note": note}}
    if as_written:
        leg["spectral_resynthesis"] = "as written (SPEC:229-232: atan2, arm_cos_f32 / arm_sin_f32 table)"
    del ch, out, iq
    torch.cuda.empty_cache()
    return leg


def engine_leg(torch, args, dev, local_rank, synth_iq):
    """The reference's own signal path (INO:53-54,81-86,198), not this build's many-channel receiver: rdsp_engine_t =
    AudioSDR::update() as the firmware image computes it (bit for bit, tests/test_engine_kat.py) and behind it the CONV
    stage with the NLMS on, at the sketch's start-up settings, native rate (44.1 kHz, no decimation: 8 B per sample of
    algorithm), 4096 receivers x 32 blocks per step.  Every stage of the engine but its Hilbert transformer is a recursion
    in time evaluated in the image's operation order, so the parallel axes are channels and rails only: the leg is bound by
    the latency of those recursions, and its HBM fraction says so.  CPU beside it: the oracle's restatement of the same
    engine, one thread, a bounded sample."""
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
    import oracle_lib
    from cases import CONV_LITERAL
    from radiodsp_sdr_rx_amd.chain import Chain
    from radiodsp_sdr_rx_amd.engine import Engine
    nch, nblk, steps = 4096, 32, 20
    n = nblk * 128
    host = synth_iq(nch, n, n_threads=host_cores(1))
    iq = torch.from_numpy(host).to(dev)
    eng = Engine(nch, max_blocks_per_call=nblk, device=local_rank, tables=oracle_lib.engine_tables())
    eng.sketch_setup()
    conv = Chain(nch, max_blocks_per_call=nblk, device=local_rank, **dict(CONV_LITERAL, lms_nr=15))
    mid, out = torch.empty_like(iq), torch.empty_like(iq)
    res = {}
    for name, both in (("engine", False), ("engine_then_conv_stage", True)):
        for _ in range(5):
            eng.update(iq, out=mid)
            if both:
                conv.process(mid, out=out)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(steps):
            eng.update(iq, out=mid)
            if both:
                conv.process(mid, out=out)
        ev[1].record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        v = float(nch) * n * steps / el
        res[name] = {"ms_per_step": el / steps * 1e3, "ms_per_step_events": ev[0].elapsed_time(ev[1]) / steps, "value": v / 1e6,
                     "unit": "IQ Msamples/s", "achieved_GBps": 8.0 * v / 1e9, "frac_of_hbm_peak": 8.0 * v / 1e9 / HBM_PEAK_GBS}
    o = oracle_lib.OracleEngine()
    x = host[0]
    t0, blocks = time.perf_counter(), 0
    while time.perf_counter() - t0 < 3.0:
        o.run(x)
        blocks += nblk
    cpu = blocks * 128 / (time.perf_counter() - t0) / 1e6
    del eng, conv, iq, mid, out
    torch.cuda.empty_cache()
    return {"config": "engine_literal",
            "workload": f"the reference's own engine: {nch} receivers x {nblk} blocks of 128 int16 IQ samples @44.1 kHz per step; AudioSDR::update() "
                        "as the firmware image computes it (LSBmode, audio2700, AGC medium; INO:117-139), then the CONV stage with the NLMS (INO:172-198)",
            "parity": "int16 audio of the engine bit for bit the image's (37 cases, tests/test_engine_kat.py); the chained path within one count "
                      "(tests/test_sketch_path.py)",
            "algorithmic_bytes_per_sample": 8.0, "channels_per_gpu": nch, "steps": steps, "legs": res,
            "bound": "latency of serial recursions (biquad cascades, oscillator phase, AGC): one lane per channel and rail; not HBM",
            "cpu_baseline": {"value": cpu, "unit": "IQ Msamples/s", "cores": 1, "kind": "port",
                             "sample": f"oracle/rdsp_engine_oracle.c, one receiver, {blocks} blocks in 3 s"}}


def main():
    args = parse()
    if os.environ.get("RDSP_BENCH_LIB"):
        sys.exit("RDSP_BENCH_LIB is no longer read: pass --lib PATH (the in-tree library would have been measured silently)")
    if args.lib:   # A/B harness: another build of the library (no torch / HIP touched by this import)
        from radiodsp_sdr_rx_amd import _lib
        _lib.use_library(args.lib)
    if args.cpu_baseline_worker:
        return f1_cpu_baseline(args) if args.config == "F1" else cpu_baseline_worker(args)
    if args.config == "F1":
        return f1_main(args)
    if args.config == "K1":
        return k1_main(args)
    if args.config == "ENGINE":   # the reference's own engine path alone (the `engine_literal` leg of the default run)
        import torch
        from radiodsp_sdr_rx_amd.chain import synth_iq
        assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
        torch.cuda.set_device(0)
        print(json.dumps(engine_leg(torch, args, torch.device("cuda", 0), 0, synth_iq)))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)   # before any torch / HIP call in this process

    import numpy as np
    import torch
    import torch.distributed as dist
    import radiodsp_sdr_rx_amd as R
    from radiodsp_sdr_rx_amd.chain import Chain, synth_iq

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on the C-level stdout; keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}"
    if args.dry_run:
        return dry_run(args, rank, world, dist)
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    if os.environ.get("RDSP_BENCH_ONE_DEVICE"):  # rehearsal of N > 1 on a one-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    kc = R.K_CONFIGS[args.config]
    cfg = dict(kc["cfg"])
    nch = args.channels_per_gpu or kc["channels"]
    nblk = args.blocks
    decim = cfg.get("decim", 4)
    n_samples = nblk * 128

    # synthetic IQ generated on the host for this rank's channel range, then resident in HBM
    threads = host_cores(1)
    t0 = time.perf_counter()
    iq_host = synth_iq(nch, n_samples, ch0=rank * nch, cw=kc.get("cw", False), n_threads=threads)
    iq = torch.from_numpy(iq_host).to(dev)
    del iq_host
    out = torch.empty((nch, n_samples // decim, 2), dtype=torch.int16, device=dev)
    gen_s = time.perf_counter() - t0

    # stage A3 in the frequency domain with 448-sample frames (rdsp_chain_set_fir_variant 2): the library's default
    # frames one granule at a time so that the bits do not depend on how a stream is cut into calls; a bench step
    # is one fixed-size call
    fir_variant = int(os.environ.get("RDSP_FIR_VARIANT", "2"))   # A/B runs: -1 = the library default, 0 = the direct form
    chain = Chain(nch, max_blocks_per_call=nblk, device=local_rank, fir_variant=fir_variant, **cfg)
    chain.set_pipelined(not args.no_pipeline)
    if args.spectral_as_written:
        chain.set_spectral_resynthesis(True)
    if args.groups > 1:
        import numpy as np
        chain.set_groups(np.arange(nch, dtype=np.uint16) % args.groups)
        for g in range(args.groups):  # distinct masks: the kernel's mask reads become a gather by group
            chain.group_reInitializeFilter(g, cfg.get("flo_hz", 300.0) + 10.0 * (g % 16),
                                           cfg.get("fhi_hz", 2700.0) - 10.0 * (g // 16 % 16))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    if os.environ.get("RDSP_PRIO"):  # A/B runs: "front_fir_prio,tail_prio"
        fpr, tpr = (int(x) for x in os.environ["RDSP_PRIO"].split(","))
        assert chain.lib.rdsp_chain_set_priorities(chain.h, fpr, tpr) == 0
    if os.environ.get("RDSP_TAIL_VARIANT"):  # A/B runs: "16" (DPP), "16m" / "8m" (matrix-pipe reduction)
        v = os.environ["RDSP_TAIL_VARIANT"]
        if v == "1step":      # round 1's tail kernel: one reduction per step (EXPERIMENTAL builds)
            chain.set_tail_variant(16, 3)
        elif v == "4step":      # round 3: four steps per reduction (EXPERIMENTAL builds)
            chain.set_tail_variant(16, 5)
        elif v == "lookahead":  # round 3: weights one block stale (EXPERIMENTAL builds)
            chain.set_tail_variant(16, 4)
        elif v.endswith("r"):   # row layouts: "16r", "8r"
            chain.set_tail_variant(int(v[:-1]), 2)
        else:
            chain.set_tail_variant(int(v.rstrip("m")), int(v.endswith("m")))
    if os.environ.get("RDSP_AUDIO_IIR"):   # A/B runs: SDR.setAudioFilter() as the 8th-order IIR bank instead of the mask
        chain.setAudioFilterKind(1)
        chain.setAudioFilter(int(os.environ["RDSP_AUDIO_IIR"]))
    if os.environ.get("RDSP_SUB_BATCH"):  # A/B runs: channels per launch in pipelined mode (0 = off)
        chain.set_sub_batch(int(os.environ["RDSP_SUB_BATCH"]))
    if os.environ.get("RDSP_FRONT_VARIANT"):  # A/B runs: force the full (0) or lean (1) front kernel
        chain.set_front_variant(int(os.environ["RDSP_FRONT_VARIANT"]))
    # set-up, un-timed: the same kernels back to back without overlap (reference durations for the
    # JSON line).  It runs before the warm-up so that the GPU has left its idle clocks by then.
    iso = None
    if not args.no_kernel_timing and not args.no_pipeline and not args.no_iso:
        chain.set_pipelined(False)
        for _ in range(20):      # the clocks of an idle GPU ramp over ~15 steps
            chain.process(iq, out=out)
        chain.set_timing(True)
        for _ in range(10):
            chain.process(iq, out=out)
        torch.cuda.synchronize()
        f2, t2, n2 = chain.get_timing()
        chain.set_timing(False)
        iso = {chain.front_kernel_name(): f2 / max(n2, 1), "rdsp_tail_kernel": t2 / max(n2, 1)}
        chain.set_pipelined(True)
    # set-up, un-timed: the package's power management needs ~15 steps of the PIPELINED load to settle (the two
    # kernels together sit at the 1400 W cap; behind the lighter set-up above the first pipelined steps run
    # 1.19, 1.18, 1.16, 1.15 ... ms before the clock finds its level at ~1.10: tests/micro/step_trace.py).  The
    # timed region is a stream in flight, not a cold start; the count goes into the line (`setup_steps`).
    settle = max(0, 40 - args.warmup)
    for _ in range(settle):
        chain.process(iq, out=out)
    for _ in range(args.warmup):
        chain.process(iq, out=out)
    chain.flush()
    barrier()
    chain.set_timing(not args.no_kernel_timing)
    t0 = time.perf_counter()
    for k in range(args.steps):
        if args.groups > 1 and args.retune_every and k % args.retune_every == 0:
            chain.group_pbt((k // args.retune_every) % args.groups, k & 1, 1 if (k >> 1) & 1 else -1)
        chain.process(iq, out=out)
    chain.flush()  # every step's audio is complete inside the timed region
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    front_ms, tail_ms, calls = chain.get_timing()
    span_ms, _ = chain.get_timing_span()
    chain.set_timing(False)
    elapsed, per_rank = gather_per_rank(torch, dist, world, elapsed, args.steps, span_ms / (calls - 1) if calls > 1 else None)

    # After the headline leg, the other GPU configurations of BASELINE.json, each timed the same way (extra_leg):
    # N = 1: K2, K4 and the K5 per-GPU shape; N > 1: the K5 shape (8192 channels per GPU, 65 536 at N = 8).  The
    # headline `value` / `config` stay the K3 per-GPU workload for every N.
    fname = chain.front_kernel_name()   # rdsp_front_fd_kernel (stage A3 in the frequency domain) or rdsp_front_kernel
    legs = {}
    default_shape = args.config == "K3" and not args.channels_per_gpu and args.groups == 1
    if default_shape and not args.no_extra_legs and not os.environ.get("RDSP_BENCH_NO_K5_LEG"):
        # K3_default: the headline workload with the library's DEFAULT decimator (one granule per frame), i.e. what a
        # caller who selects nothing gets; then the other configurations in the headline's form
        which = ["K5"] if world > 1 else ["K3_default", "K3_as_written", "K2", "K2_default", "K5", "K4", "engine_literal"]
        for name in which:
            reuse = iq if name in ("K2", "K2_default", "K3_default", "K3_as_written") else None   # same generator, same channels as the headline leg
            try:
                if name == "K2_default":   # K2 with the library's default decimator: no tail stage follows, so the row form
                    # of round 6 (rdsp_front_rd_kernel; split-invariant like the one-granule form K3_default runs)
                    legs[name] = extra_leg("K2", torch, dist, R, Chain, synth_iq, args, rank, world, local_rank, dev,
                                           -1, barrier, iq=reuse, label=name)
                    continue
                if name == "K3_default":
                    legs[name] = extra_leg("K3", torch, dist, R, Chain, synth_iq, args, rank, world, local_rank, dev, -1,
                                           barrier, iq=reuse, label="K3_default")
                    continue
                if name == "K3_as_written":   # the headline workload with the spectral stage's re-synthesis as SPEC:229-232 writes it
                    legs[name] = extra_leg("K3", torch, dist, R, Chain, synth_iq, args, rank, world, local_rank, dev, fir_variant,
                                           barrier, iq=reuse, label="K3_as_written", as_written=True)
                    continue
                if name == "engine_literal":
                    legs[name] = engine_leg(torch, args, dev, local_rank, synth_iq)
                    continue
                legs[name] = extra_leg(name, torch, dist, R, Chain, synth_iq, args, rank, world, local_rank, dev, fir_variant,
                                       barrier, iq=reuse)
            except Exception as e:   # an extra leg never takes the headline down
                if world > 1:
                    raise
                legs[name] = {"config": name, "value": None, "note": f"failed: {e}"}
    k5_leg = None
    if world > 1 and "K5" in legs:
        k5_leg = dict(legs["K5"], note="BASELINE.json configs[4]: full chain, 8192 channels per GPU, timed after the headline "
                                       "leg with the same barrier / max-over-ranks protocol")

    if rank == 0:
        total_samples = float(world) * nch * n_samples * args.steps
        value = total_samples / elapsed / 1e6  # IQ Msamples/s, whole job
        B = algorithmic_bytes_per_sample(decim)
        # dominant kernel of the step and its roofline (bytes per launch / avg launch time)
        front_avg = front_ms / max(calls, 1)
        tail_avg = tail_ms / max(calls, 1)
        # dominant kernel: whichever runs longer (in pipelined mode both run for the whole step, within a few percent)
        dom, dom_ms, B_own = dominant_kernel(fname, front_avg, tail_avg, decim)
        # roofline.achieved: SURVEY 8(d)'s figure (B = 5 bytes per input IQ sample of the chain) x the
        # samples one launch processes / the dominant kernel's average duration; the kernel's own
        # algorithmic bytes (the tail kernel only moves 4 B in + 4 B out per OUTPUT sample) go beside it
        bytes_per_launch = B * nch * n_samples
        achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        own = B_own * nch * n_samples / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # counters of a committed PMC pass of this configuration (tests/profile_round.sh ->
        # profiles/counters.json): HBM bytes and VALU busy fraction per launch of each kernel
        ctr, ctr_note = load_counters(args.config)
        if (nch, nblk) != (kc["channels"], 512):
            ctr, ctr_note = {}, "PMC constants are for the default shape of the configuration only"
        traffic = ctr.get(dom, {}).get("hbm_bytes")
        chain_traffic = sum(v.get("hbm_bytes", 0.0) for k, v in ctr.items() if k in (fname, "rdsp_tail_kernel")) or None
        flops = FLOP_PER_SAMPLE.get(args.config, 0) * float(nch) * n_samples   # per launch of the chain
        achieved_tf = flops / (elapsed / args.steps) / 1e12
        res = {
            "metric": "IQ Msamples/s through full SSB+NR chain; achieved HBM GB/s vs peak",
            "value": value,
            "unit": "IQ Msamples/s",
            "n_gpus": world,
            "devices_visible": torch.cuda.device_count(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # what a long stream pays per step: the period between the ends of consecutive steps' last kernels
            # (HIP events), i.e. without the pipeline's fill -- the first step's front kernel has no tail kernel
            # to overlap with, ~0.6 ms once per timed region, 0.03 ms per step at --steps 20
            "ms_per_step_steady": span_ms / (calls - 1) if calls > 1 else None,
            "per_rank_ms": per_rank,   # every rank's own wall time per step and steady-state period (value uses the max)
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: {nch} channels/GPU x {nblk} blocks of 128 int16 IQ samples @96 kHz per step; "
                            + WORKLOAD_TEXT[args.config]
                            + (f" + audio filter {os.environ['RDSP_AUDIO_IIR']} as the 8th-order IIR bank (non-default option)"
                               if os.environ.get("RDSP_AUDIO_IIR") else ""),
                "channels_per_gpu": nch,
                "blocks_per_step": nblk,
                "sharding": f"channels x{world}, no collectives",
                "pipelined": not args.no_pipeline,
                "decimator": DECIMATOR_TEXT.get(fir_variant, DECIMATOR_TEXT[-1]),
                "spectral_resynthesis": ("as written (SPEC:229-232: atan2, arm_cos_f32 / arm_sin_f32 table)" if args.spectral_as_written
                                         else "X mag'/mag (exact-arithmetic equivalent of SPEC:229-232; the as-written form sits 1.7e-5..1.9e-5 away)"),
                "groups": args.groups,
                "retune_every_steps": args.retune_every,
                # A/B switches read from the environment by measurement scripts; a driver run shows an empty dict
                "ab_switches": {k: v for k, v in os.environ.items() if k.startswith("RDSP_")},
            },
            "chain_hbm": {"algorithmic_bytes_per_sample": B, "achieved_GBps": B * value * 1e6 / 1e9 / world,
                          "frac_of_peak": B * value * 1e6 / 1e9 / world / HBM_PEAK_GBS,
                          "traffic_bytes_per_step": chain_traffic},
            "kernels_ms_per_step": {fname: front_avg, "rdsp_tail_kernel": tail_avg},
            "kernels_ms_isolated": iso,
            # The north-star's roofline is HBM, and that is what `roofline` reports (algorithmic bytes of
            # the dominant kernel / its measured duration / 8 TB/s).  What binds first is the fp32 VALU
            # (SURVEY 0.7, Appendix D): `limiter` says so and `valu` carries that roofline -- the flop model
            # of the chain as run / step time against the 157.3 TFLOP/s vector peak, and the VALU-busy
            # fraction of the dominant kernel from the committed PMC pass.
            "roofline": {"bound": "hbm", "limiter": "valu", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "counters_note": ctr_note,
                         "kernel_own": {"algorithmic_bytes_per_sample": B_own, "achieved": own, "frac": own / HBM_PEAK_GBS},
                         "valu": {"achieved": achieved_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": achieved_tf / FP32_PEAK_TFLOPS,
                                  "flop_per_sample_model": FLOP_PER_SAMPLE.get(args.config),
                                  "busy_frac_pmc": {k: v.get("valu_busy_frac") for k, v in ctr.items()
                                                    if k in (fname, "rdsp_tail_kernel")} or None}},
            "input_gen_s": gen_s,
            "setup_steps": {"unpipelined_kernel_reference": 30 if iso is not None else 0, "settle_before_warmup": settle,
                            "note": "un-timed, before the --warmup steps: clock / power settling of the GPU under this load"},
            # what "matches the reference's CPU path" means here (tests/parity_util.py; a deviation from the
            # north-star's blanket 1e-5 for the bare-recursion chains, stated where the number is read)
            "tolerance": {"int16_unpack_pack": "bit-exact",
                          "feed_forward_chains_and_K3": "<= 1e-5 normwise per channel vs the float32 CPU oracle (K3 measured 3e-6..6e-6), "
                                                        "the spectral stage's re-synthesis in the form named in config.spectral_resynthesis on both sides",
                          "spectral_resynthesis_forms": "SPEC:229-232 as written (atan2 + CMSIS table sine / cosine) and the exact-arithmetic "
                                                        "equivalent X mag'/mag sit 1.7e-5..1.9e-5 of the peak apart (the table's interpolation error); "
                                                        "each GPU form is tested <= 1e-5 against the oracle evaluating the SAME form; the as-written "
                                                        "form costs +26 % per K3 step (--spectral-as-written)",
                          "integer_analysers": "bit-exact; windows, q15 twiddles and sqrt guess table equal the reference's firmware image",
                          "bare_recursion_chains": "DSP-NR / ALS without spectral stage + AGC, IIR bank, SAM: truth-anchored -- "
                                                   "err(gpu, float64 model) <= max(1e-5, 1.5 x err(oracle, float64 model)); gpu vs "
                                                   "oracle up to 1.4e-4 there, the float32 oracle itself 4e-5..1.1e-4 from float64",
                          "oracle": "pinned on the reference's own compiled code where its shipped build has the stage (firmware routines run "
                                    "under an instruction-set interpreter): the CONV stage with NLMS at the ONE shape the image links -- FFT_L 256, "
                                    "decim 1, 44.1 kHz, 129 taps -- 1.6e-7..2.0e-6 oracle, 1.5e-7..2.1e-6 GPU (FFT_L 512 .. 4096 are other "
                                    "instantiations checked against the oracle only); analysers, AudioFilterBiquad, arm_lms_norm_f32 bit for bit; "
                                    "and since round 6 the reference's whole engine (AudioSDR::update: mixer, side-band selection, AM / SAM, audio "
                                    "filters, hang AGC, ALS, blanker; AudioSDRpreProcessor) bit for bit as rdsp_engine_t / rdsp_preproc_t "
                                    "(configs.engine_literal).  THIS line's workload, K3, runs rdsp_chain_t: a 96 kHz many-channel chain with a "
                                    "decimator and a spectral stage the reference has no compiled counterpart of -- its NCO / decimator / "
                                    "spectral stage / ALS / AGC are this build's designs, checked against the CPU oracle only (DESIGN.md 2)"},
        }
        if args.lib:
            res["library"] = os.path.abspath(args.lib)   # an A/B run, not the in-tree build
        if legs:
            res["configs"] = legs      # K2 / K4 / K5-per-GPU legs of the same run (N > 1: K5 only)
        if k5_leg is not None:
            res["k5"] = k5_leg
        if world == 1 and not args.no_host_io:
            # extra, un-timed for `value`: the same chain fed from and drained to host memory through
            # rdsp_stream_run_memory (pinned double buffers, upload / kernels / download overlapped)
            try:
                from radiodsp_sdr_rx_amd.io import stream_memory
                h_blocks, h_per = 256, 64
                h_np = synth_iq(nch, h_blocks * 128, cw=kc.get("cw", False), n_threads=threads)
                chain.set_pipelined(not args.no_pipeline)
                legs = {}
                for kind in ("pageable", "pinned"):
                    if kind == "pinned":   # page-locked at both ends: no staging copies
                        h_iq = torch.from_numpy(h_np).pin_memory()
                        h_out = torch.zeros((nch, h_blocks * 128 // decim, 2), dtype=torch.int16).pin_memory()
                        warm = h_iq[:, :h_per * 2 * 128].contiguous().pin_memory()
                    else:
                        import numpy as np
                        h_iq, h_out = h_np, np.zeros((nch, h_blocks * 128 // decim, 2), np.int16)
                        warm = h_np[:, :h_per * 2 * 128]
                    chain.reset()
                    stream_memory(chain, warm, h_per)  # warm-up
                    chain.reset()
                    _, st = stream_memory(chain, h_iq, h_per, out=h_out)
                    tot = float(nch) * st["samples_in"]
                    legs[kind] = {"value": tot / st["seconds"] / 1e6,
                                  "pcie_GBps": (tot * 4 + float(nch) * st["samples_out"] * 4) / st["seconds"] / 1e9,
                                  "seconds": st["seconds"]}
                    del h_iq, h_out, warm
                res["host_io"] = {"value": legs["pinned"]["value"], "unit": "IQ Msamples/s",
                                  "pcie_GBps": legs["pinned"]["pcie_GBps"], "pageable": legs["pageable"],
                                  "sample": f"{nch} channels x {h_blocks} blocks through rdsp_stream_run_memory, "
                                            f"{h_per} blocks per batch, host arrays page-locked (value) and pageable; "
                                            f"device buffer set-up inside the timed run"}
                del h_np
            except Exception as e:
                res["host_io"] = {"value": None, "sample": f"failed: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker",
                                    "--config", args.config, "--blocks", str(nblk),
                                    "--cpu-seconds", str(args.cpu_seconds)],
                                   capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
                res["cpu_baseline"] = json.loads(line)
            except Exception as e:  # the baseline is a reported number, never the target
                res["cpu_baseline"] = {"value": None, "unit": "IQ Msamples/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {e}"}
            try:   # for scale only: what the reference's own processor executes (counted by the interpreter that made the fixture)
                kat = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "firmware_kat.npz"))
                res["cpu_baseline"]["reference_on_its_own_hardware"] = {
                    "instructions_per_128_sample_block": {"conv_stage": int(kat["conv_plain_instructions_per_block"]),
                                                          "conv_stage_with_nlms": int(kat["conv_nr15_instructions_per_block"])},
                    "note": "the firmware image's doConvolutionalProcessing (one receiver, no decimation / demodulation / AGC: "
                            "the CONV stage only) counted under tests/golden/thumb_emu.py; a 600 MHz Cortex-M7 issues at most "
                            "two per cycle; not a timing, not comparable with `value`"}
            except Exception:
                pass
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
