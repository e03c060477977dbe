"""CPU tests of the product's host side: the C-ABI library loads and exports every
symbol include/rdsp.h declares, design helpers agree with the oracle, the synthetic
generator is deterministic and shardable, host emulation of the kernels' index
arithmetic passes, and the product fails loudly without a GPU.  No compute calls."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(rdsp):
    hdr = open(os.path.join(ROOT, "include", "rdsp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(rdsp_[A-Za-z0-9_]+)\s*\(", hdr))
    assert len(names) > 45
    lib = C.CDLL(os.path.join(ROOT, "radiodsp_sdr_rx_amd", "librdsp_hip.so"))
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing
    from radiodsp_sdr_rx_amd import _lib
    bound = {s[0] for s in _lib.SYMBOLS}
    assert names <= bound, sorted(names - bound)


def test_config_struct_layout_matches_oracle(rdsp, oracle):
    from radiodsp_sdr_rx_amd._lib import ChainConfig
    a = [(n, t) for n, t in ChainConfig._fields_]
    b = [(n, t) for n, t in oracle.OrcConfig._fields_]
    assert a == b and C.sizeof(ChainConfig) == C.sizeof(oracle.OrcConfig) == 104


def test_product_fails_loudly_without_gpu(rdsp):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from radiodsp_sdr_rx_amd import RdspError
    from radiodsp_sdr_rx_amd.chain import Chain
    with pytest.raises(RdspError) as e:
        Chain(4)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "radiodsp_sdr_rx_amd")):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)


@pytest.mark.parametrize("window", [1, 2, 3, 4, 5])   # FIR_filter_window ids of CONV:152-179 (5 = the default branch)
@pytest.mark.parametrize("fft_l,lo,hi,fs", [(256, 300.0, 4000.0, 44117.64706), (512, 300.0, 2700.0, 24000.0),
                                           (4096, 450.0, 950.0, 24000.0), (1024, -2700.0, -300.0, 24000.0)])
def test_design_helpers_match_oracle(rdsp, oracle, fft_l, lo, hi, fs, window):
    from radiodsp_sdr_rx_amd.chain import calc_cplx_FIR_coeffs, init_filter_mask
    ci, cq = calc_cplx_FIR_coeffs(fft_l // 2 + 1, lo, hi, fs, window)
    lib = oracle.load()
    oi, oq = np.zeros_like(ci), np.zeros_like(cq)
    lib.orc_calc_cplx_FIR_coeffs(oi.ctypes.data_as(C.POINTER(C.c_double)), oq.ctypes.data_as(C.POINTER(C.c_double)),
                                 len(ci), lo, hi, fs, window)
    assert np.abs(ci - oi).max() < 1e-15 and np.abs(cq - oq).max() < 1e-15
    m = init_filter_mask(ci, cq, fft_l)
    om = np.zeros(2 * fft_l, np.float32)
    lib.orc_init_filter_mask(om.ctypes.data_as(C.POINTER(C.c_float)), oi.ctypes.data_as(C.POINTER(C.c_double)),
                             oq.ctypes.data_as(C.POINTER(C.c_double)), fft_l)
    # float64 DFT (product) vs float32 FFT (oracle) of the same float-narrowed taps
    assert np.abs(m - om).max() < 2e-6


def test_synth_is_deterministic_and_shardable(rdsp):
    from radiodsp_sdr_rx_amd.chain import synth_iq
    a = synth_iq(6, 4096)
    b = synth_iq(6, 4096, n_threads=3)
    assert np.array_equal(a, b)
    # any (channel, time) window can be generated independently
    part = synth_iq(2, 1024, ch0=3, t0=2048)
    assert np.array_equal(part, a[3:5, 2048:3072])
    x = (a[..., 0] + 1j * a[..., 1]) / 32767.0
    assert 0.3 < np.sqrt((np.abs(x) ** 2).mean()) < 0.6 and np.abs(a).max() < 32767
    # the two USB tones and the interferer sit at f_off + {700, 1900, 1000} Hz
    X = np.abs(np.fft.fft(x[0] * np.hanning(4096)))
    binhz = 96000 / 4096
    for f, amp in ((12700, 0.2), (13000, 0.3), (13900, 0.2)):
        k = int(round(f / binhz))
        peak = X[k - 1:k + 2].max() / (4096 / 2)  # Hann coherent gain 0.5
        assert 0.7 * amp < peak < 1.2 * amp, (f, peak)
    assert X[int(round(20000 / binhz))] / 2048 < 0.02  # elsewhere only noise
    cw = synth_iq(1, 96000 // 4, cw=True)[0]
    env = np.abs(cw[:, 0] + 1j * cw[:, 1]).reshape(-1, 480).mean(axis=1)
    assert env[:10].mean() > 2 * env[14:22].mean()  # keyed 60 ms on / 60 ms off


def test_k_configs_cover_baseline_json(rdsp):
    import json
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5
    K = rdsp.K_CONFIGS
    assert K["K2"]["channels"] == 4096 and K["K3"]["channels"] == 4096 and K["K4"]["channels"] == 8192
    assert K["K3"]["cfg"]["fft_l"] == 512 and K["K4"]["cfg"]["fft_l"] == 4096 and K["K5"]["channels"] == 8192


@pytest.mark.parametrize("src", ["host_fft_check.cpp", "host_fir_check.cpp"])
def test_kernel_index_arithmetic_on_host(src, tmp_path):
    """The FFT passes / LDS layout / FIR lane code are __host__ __device__: run the
    same source thread-by-thread on the CPU against float64 references."""
    exe = str(tmp_path / src.replace(".cpp", ""))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "--offload-arch=gfx950", "-I",
                           os.path.join(ROOT, "radiodsp_sdr_rx_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", src), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr


def test_fft_lds_maps_are_bank_conflict_free_under_the_lds_model(tmp_path):
    """The per-exchange LDS maps compiled into FftPlan (printed by the host harness) against the bank model
    of tests/micro/lds_model.py (ds_write_b64: 16 contiguous lanes over 32 banks; ds_read_b64: 32 lanes
    over 64 banks): every exchange of the one-wave plans costs exactly the conflict-free cycles for both of
    its lane patterns, as written and as read, and fits the work buffer; the four-wave plans keep phi."""
    import importlib.util
    exe = str(tmp_path / "host_fft_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-std=c++17", "--offload-arch=gfx950", "-I",
                           os.path.join(ROOT, "radiodsp_sdr_rx_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host", "host_fft_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "maps"], capture_output=True, text=True, check=True).stdout
    spec = importlib.util.spec_from_file_location("lds_model", os.path.join(ROOT, "tests", "micro", "lds_model.py"))
    lm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lm)
    seen = 0
    for line in out.splitlines():
        f = line.split()
        if not f or f[0] != "map":
            continue
        N, P, x, shift, mul, wb = (int(v) for v in f[1:])
        A = lambda i: i + mul * (i >> shift)
        assert A(N - 1) + 1 <= wb, (N, P, x)
        nt, pats = lm.plan(N, P)
        if nt == 64:
            for pat in (pats[x], pats[x + 1]):
                assert lm.pattern_cost(N, P, A, pat) == (1.0, 1.0), (N, P, x)
            assert (shift, mul) == lm.PERX[(N, P)][x]
        else:
            assert (1 << shift, mul) == (P, 1)
        seen += 1
    assert seen == 3 + 2 + 2 + 3 + 2


def _device_disassembly(tmp_path):
    """gfx950 disassembly of every code object linked into librdsp_hip.so, by kernel name"""
    llvm = "/opt/rocm/lib/llvm/bin"
    lib = os.path.join(ROOT, "radiodsp_sdr_rx_amd", "librdsp_hip.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    sec = str(tmp_path / "fatbin.bin")
    subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + sec, lib, str(tmp_path / "unused.so")])
    blob = open(sec, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in __import__("re").finditer(magic, blob)]
    kernels = {}
    for n, a in enumerate(starts):
        b = starts[n + 1] if n + 1 < len(starts) else len(blob)
        part, co = str(tmp_path / f"b{n}.bin"), str(tmp_path / f"b{n}.co")
        open(part, "wb").write(blob[a:b])
        subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
        name = None
        for line in subprocess.run([llvm + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True,
                                   text=True, check=True).stdout.splitlines():
            if line[:1].isdigit() and line.rstrip().endswith(">:"):
                name = line.split("<", 1)[1][:-2]
                kernels[name] = []
            elif name and line.startswith("\t"):
                kernels[name].append(line.split()[0])
    return kernels


def test_generated_code_keeps_the_instruction_forms_the_measurements_rest_on(tmp_path):
    """What round 3 found in the generated code, pinned (DESIGN.md 4.1 'LDS', 'No barriers inside a wave'): a
    compiler that goes back to any of these forms costs 2-13 % per step without failing a parity test.
    Frequency-domain front kernels: no FLAT loads in the frame loop (at most the one pointer select of the PRE
    prologue), no ds_read2_b64 (LDS reads are single ds_read_b64), input quads through buffer loads; the
    one-wave kernels contain no s_barrier at all."""
    k = _device_disassembly(tmp_path)
    fd = {n: v for n, v in k.items() if "rdsp_front_fd_kernel" in n}
    assert len(fd) >= 16
    for n, ins in fd.items():
        assert ins.count("flat_load_dwordx2") == 0 and sum(i.startswith("flat_load") for i in ins) <= 1, n
        assert "ds_read2_b64" not in ins and ins.count("ds_read_b64") >= 16, n
        assert ins.count("buffer_load_dwordx4") >= 7, n
        one_wave = any(t in n for t in ("ILi256ELi4E", "ILi512ELi8E", "ILi1024ELi16E"))
        assert ("s_barrier" in ins) != one_wave, n
    rd = {n: v for n, v in k.items() if "rdsp_front_rd_kernel" in n}   # the decimator on 16-lane rows: the same forms
    assert len(rd) >= 12
    for n, ins in rd.items():
        assert not any(i.startswith("flat_load") for i in ins) and "ds_read2_b64" not in ins, n
        assert ins.count("buffer_load_dwordx4") >= 24, n                # twelve quads a lane: prologue and prefetch
        one_wave = any(t in n for t in ("ILi256ELi4E", "ILi512ELi8E", "ILi1024ELi16E"))
        assert ("s_barrier" in ins) != one_wave, n
    for name in ("rdsp_tail_kernel", "rdsp_tail_dual_kernel", "rdsp_spectrum_kernel", "rdsp_fft1024_kernel",
                 "rdsp_biquad_kernel", "rdsp_sam_kernel"):
        hit = [v for n, v in k.items() if name in n]
        assert hit and all("s_barrier" not in v for v in hit), name
    tail = [v for n, v in k.items() if "rdsp_tail_kernel" in n][0]
    assert tail.count("ds_read_b64") >= 32   # the even sample pairs
    # arm_biquad_cascade_df1_f32 rounds every product before it adds it (the reference's image: 5 VMUL + 4 VADD per
    # sample); hipcc contracts by default, so the cascade kernel must show no fused operation at all
    bq = [v for n, v in k.items() if "rdsp_biquad_kernel" in n][0]
    assert not any(i.startswith(("v_fma", "v_fmac", "v_mad_f32", "v_mac_f32", "v_pk_fma")) for i in bq)
    assert sum(i.startswith("v_mul_f32") for i in bq) >= 45 and sum(i.startswith("v_add_f32") for i in bq) >= 36


def test_register_budget_of_the_two_kernels_that_share_a_simd(tmp_path):
    """Pipelined K3 keeps two waves of the frequency-domain front kernel and one tail wave on a SIMD: 2 x 176 +
    128 of its 512 registers.  Nothing in the source enforces that (an `amdgpu_num_vgpr`
    attribute on the tail kernel turned out to be inert): read the counts out of the built code objects."""
    llvm = "/opt/rocm/lib/llvm/bin"
    _device_disassembly(tmp_path)            # leaves the unbundled code objects b<n>.co in tmp_path
    vg, sspill = {}, {}
    for co in sorted(tmp_path.glob("b*.co")):
        name = None
        for line in subprocess.run([llvm + "/llvm-readelf", "--notes", str(co)], capture_output=True, text=True,
                                   check=True).stdout.splitlines():
            t = line.strip()
            if t.startswith(".name:"):
                name = t.split(":", 1)[1].strip()
            elif t.startswith(".vgpr_count:") and name:
                vg[name] = int(t.split(":", 1)[1])
            elif t.startswith(".sgpr_spill_count:") and name:
                sspill[name] = int(t.split(":", 1)[1])
    tail = [v for n, v in vg.items() if "rdsp_tail_kernel" in n]
    front = [v for n, v in vg.items() if "rdsp_front_fd_kernel" in n and "ILi512ELi8ELb0ELb0E" in n]
    assert tail and front, sorted(vg)[:5]
    alloc = lambda v: (v + 7) // 8 * 8        # gfx950 allocates VGPRs in blocks of 8
    assert alloc(tail[0]) <= 128 and alloc(front[0]) <= 176 and 2 * alloc(front[0]) + alloc(tail[0]) <= 512, (tail, front)
    # scalar registers spilled to VGPR lanes come back as v_readlane in the frame loop, each a vector issue slot (DESIGN.md
    # 8.3c: 79 -> 34 in the K2 instance once cold kernel parameters were read where they are used, -1.9 % per step)
    k2 = [v for n, v in sspill.items() if "rdsp_front_fd_kernel" in n and "ILi256ELi4ELb0ELb0ELb1ELi7E" in n]
    k3 = [v for n, v in sspill.items() if "rdsp_front_fd_kernel" in n and "ILi512ELi8ELb0ELb0ELb0ELi7E" in n]
    assert k2 and k3 and k2[0] <= 48 and k3[0] <= 24, (k2, k3)


def test_branch_spectra_of_the_row_form_are_the_transforms_of_the_polyphase_taps():
    """rdsp_rd_decimator_image (rdsp_design.c): [r][bin] = DFT_256 of g_r[k] = h[4k - r] (k <= 64), divided by 256 -- what
    rdsp_front_rd_kernel multiplies the four low-rate transforms with.  Checked against numpy's FFT for a random tap
    set, and the overlap-save identity behind it on a random stream: windows of 64 + 128 quads, the spectra, the
    window's outputs 64..191 are the direct 256-tap decimating convolution."""
    lib = C.CDLL(os.path.join(ROOT, "radiodsp_sdr_rx_amd", "librdsp_hip.so"))
    f32p = C.POINTER(C.c_float)
    lib.rdsp_rd_decimator_image.argtypes = [f32p, f32p]
    lib.rdsp_rd_decimator_image.restype = C.c_int
    rng = np.random.default_rng(3)
    h = rng.standard_normal(256).astype(np.float32)
    img = np.zeros(2 * 4 * 256, np.float32)
    assert lib.rdsp_rd_decimator_image(h.ctypes.data_as(f32p), img.ctypes.data_as(f32p)) == 0
    G = (img[0::2] + 1j * img[1::2]).reshape(4, 256)
    for r in range(4):
        g = np.zeros(256)
        for k in range(65):
            if 0 <= 4 * k - r < 256:
                g[k] = h[4 * k - r]
        assert np.abs(G[r] - np.fft.fft(g) / 256).max() <= 2e-7 * np.abs(np.fft.fft(g) / 256).max()
    x = rng.standard_normal(4 * 192) + 1j * rng.standard_normal(4 * 192)         # 64 history quads and one frame
    win = np.zeros((4, 256), complex)
    for r in range(4):
        win[r, :192] = x[r::4]
    y = np.fft.ifft(sum(np.fft.fft(win[r]) * G[r] * 256 for r in range(4)))
    ref = np.array([sum(h[k] * x[4 * m - k] for k in range(256)) for m in range(64, 192)])
    assert np.abs(y[64:192] - ref).max() <= 1e-5 * np.abs(ref).max()


def test_quad_frame_mask_gather_matches_the_radix4_device_image(rdsp):
    """front_frame_quad (FFT_L 256, four overlap-save frames per pass, 16 points per lane) reads its mask out of
    the device image of the radix-4 plan, which both front kernels share: bin k = i + 16 e of lane i, element e sits
    at 64 (e >> 2) + (e & 3) + 16 (i & 3) + 4 (i >> 2).  Checked against rdsp_mask_device_image for a labelled mask."""
    lib = C.CDLL(os.path.join(ROOT, "radiodsp_sdr_rx_amd", "librdsp_hip.so"))
    n = 256
    nat = np.zeros(2 * n, np.float32)
    nat[0::2] = np.arange(n) * n          # the image divides by N: entry k reads back as k
    nat[1::2] = -np.arange(n) * n
    img = np.zeros(2 * n, np.float32)
    f32p = C.POINTER(C.c_float)
    lib.rdsp_mask_device_image.argtypes = [f32p, C.c_int, f32p]
    lib.rdsp_mask_device_image.restype = None
    lib.rdsp_mask_device_image(nat.ctypes.data_as(f32p), n, img.ctypes.data_as(f32p))
    for i in range(16):
        for e in range(16):
            idx = 64 * (e >> 2) + (e & 3) + 16 * (i & 3) + 4 * (i >> 2)
            assert img[2 * idx] == i + 16 * e and img[2 * idx + 1] == -(i + 16 * e), (i, e, idx)


def test_host_c_under_address_and_ub_sanitizers(tmp_path):
    """rdsp_graph.c, rdsp_io.c and rdsp_design.c (no HIP in them) built with ASan + UBSan + LSan
    and walked by tests/host/host_sanitize.c: pool exhaustion, teardown with blocks queued,
    damaged WAV headers, every design routine at every size with exactly sized buffers."""
    exe = str(tmp_path / "host_sanitize")
    csrc = os.path.join(ROOT, "radiodsp_sdr_rx_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fopenmp",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "host", "host_sanitize.c")]
                          + [os.path.join(csrc, f) for f in ("rdsp_graph.c", "rdsp_io.c", "rdsp_design.c", "rdsp_q15_tables.c")]
                          + ["-lm", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "host_sanitize OK" in out.stdout, out.stdout + out.stderr
    assert "runtime error" not in out.stderr and "Sanitizer" not in out.stderr, out.stderr


def test_bin_of_pos_is_a_permutation(rdsp):
    lib = C.CDLL(os.path.join(ROOT, "radiodsp_sdr_rx_amd", "librdsp_hip.so"))
    for n in (256, 512, 1024, 2048, 4096):
        ks = sorted(lib.rdsp_bin_of_pos(n, i) for i in range(n))
        assert ks == list(range(n))


WORKER = r"""
import os, sys, numpy as np, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from radiodsp_sdr_rx_amd.chain import synth_iq
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
nch = 3
mine = synth_iq(nch, 2048, ch0=rank * nch)   # rank-local channel shard, no exchange
import torch
t = torch.tensor([float(mine.astype(np.int64).sum())], dtype=torch.float64)
dist.all_reduce(t)                            # only the test's checksum crosses ranks
full = synth_iq(nch * world, 2048)
assert np.array_equal(mine, full[rank * nch:(rank + 1) * nch])
assert float(t.item()) == float(full.astype(np.int64).sum())
tt = torch.tensor([1.0 + rank]); dist.all_reduce(tt, op=dist.ReduceOp.MAX); assert tt.item() == world
dist.barrier(); dist.destroy_process_group(); open(os.path.join(sys.argv[2], f"ok_{rank}"), "w").write("ok")
"""


def test_channel_sharding_world_size_2_gloo(tmp_path):
    """N > 1 path of bench.py: ranks own contiguous channel ranges, nothing but the
    timing barrier / max-reduce crosses ranks (gloo, CPU)."""
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(w), ROOT, str(tmp_path)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_bench_gpus_2_launches_two_ranks(tmp_path):
    """`python bench.py --gpus 2` (the driver's command form, no WORLD_SIZE in the environment)
    starts two ranks as a child process and relays rank 0's single JSON line with n_gpus = 2.
    --dry-run replaces the device work by a sleep: launcher, rendezvous (gloo, 127.0.0.1), the
    barrier + max-over-ranks reduction and the one-line protocol are the real ones."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "0",
                        "--dry-run"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["data"] == "dry-run"
    assert d["ms_per_step"] >= 2.0                     # the MAX over ranks: rank 1 sleeps 2 ms per step
    # ... and every rank's own time beside it (gloo all_gather), so that a curve below N x can be attributed
    per = d["per_rank_ms"]
    assert len(per["ms_per_step"]) == 2 and 1.0 <= per["ms_per_step"][0] < per["ms_per_step"][1]
    assert max(per["ms_per_step"]) == pytest.approx(d["ms_per_step"]) and per["ms_per_step_steady"] == [1.0, 2.0]
    # N > 1 with the default workload carries the second leg at BASELINE.json configs[4]'s shape
    assert d["k5"]["channels_per_gpu"] == 8192 and d["k5"]["channels_total"] == 16384 and d["k5"]["unit"] == d["unit"]
    # a worker whose world size disagrees with --gpus refuses to run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                         text=True, env=dict(env, WORLD_SIZE="1", RANK="0"), timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


def test_bench_quotes_pmc_constants_only_for_the_sources_they_were_taken_on(tmp_path, monkeypatch):
    """profiles/counters.json carries, per configuration, the hash of the library sources its PMC passes
    ran on; bench.load_counters returns nothing (and says why) when the tree's sources differ."""
    import importlib
    import json
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    sha = bench.lib_sha()
    assert len(sha) == 16 and sha == bench.lib_sha()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "lib_sha", lambda: sha)
    (prof / "counters.json").write_text(json.dumps({"K3": {"round": "r03", "lib_sha": sha, "rdsp_tail_kernel": {"hbm_bytes": 5.0}},
                                                    "K2": {"round": "r02", "rdsp_front_fd_kernel": {"hbm_bytes": 1.0}}}))
    ctr, note = bench.load_counters("K3")
    assert note is None and ctr == {"rdsp_tail_kernel": {"hbm_bytes": 5.0}}
    ctr, note = bench.load_counters("K2")               # no hash recorded: stale by definition
    assert ctr == {} and "not quoted" in note
    ctr, note = bench.load_counters("K4")
    assert ctr == {} and note


def test_bench_k1_line_on_the_host(tmp_path):
    """`bench.py --config K1`: BASELINE.json configs[0], the CPU reference path through the block graph"""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "K1", "--blocks", "64", "--steps", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 0 and d["value"] > 0 and d["cpu_baseline"]["cores"] == 1 and d["roofline"] is None
    assert d["audio_blocks_out"] == 16 and d["config"]["workload"].startswith("K1")


def test_as_written_resynthesis_in_closed_form_is_the_tables_own_arithmetic():
    """SPEC:229-232 multiplies the new magnitude by arm_cos_f32(phi) and arm_sin_f32(phi): CMSIS' 513-entry table with
    linear interpolation.  csrc/rdsp_kernels.hip (spec_table_factor) evaluates that interpolation in closed form: the table
    value is the exact sine / cosine times A(f) = 1 - (h^2 / 2) f (1 - f), h = 2 pi / 512, f the fraction between nodes,
    with f from a degree-11 arctangent of min / max (f (1 - f) is the same in all eight octants).  Here, over the whole
    circle and in float32 like the kernel: the closed form against the table's own arithmetic (the published routine,
    restated) -- 1e-7 of the bin where the two FORMS of the stage are 1.9e-5 apart."""
    k = np.arange(513)
    tab = np.float32(np.round(np.sin(2 * np.pi * k / 512), 8))                   # sinTable_f32's literals
    rng = np.random.default_rng(5)
    phi = np.concatenate([rng.uniform(-np.pi, np.pi, 400000), np.arange(-512, 513) * (np.pi / 512), [0.0, np.pi, -np.pi]])
    r = rng.uniform(1e-6, 3.0, len(phi))
    x, y = (r * np.cos(phi)).astype(np.float32), (r * np.sin(phi)).astype(np.float32)

    def table(turns):                                                            # arm_sin_f32, in float32
        t = np.float32(turns)
        n = np.floor(t)
        t = t - n
        fi = np.float32(512.0) * t
        idx = fi.astype(np.int32)
        wrap = idx >= 512
        idx = np.where(wrap, 0, idx)
        fi = np.where(wrap, fi - np.float32(512.0), fi)
        fr = fi - idx.astype(np.float32)
        return (np.float32(1.0) - fr) * tab[idx] + fr * tab[idx + 1]
    turns = (np.arctan2(y, x) * np.float32(0.159154943092)).astype(np.float32)
    want_c, want_s = table(turns + np.float32(0.25)).astype(np.float64), table(turns).astype(np.float64)
    F = np.float32
    ax, ay = np.abs(x), np.abs(y)
    mx, mn = np.maximum(np.maximum(ax, ay), F(1e-30)), np.minimum(ax, ay)
    z = mn * (F(1.0) / mx)
    s = z * z
    u = F(-0.954960883)
    for c in (4.29009151, -9.48728275, 15.7710886, -27.1045456, 81.4854736):
        u = u * s + F(c)
    u = u * z
    f = u - np.floor(u)
    a = (F(1.0) + (f - f * f) * F(-7.52982e-05)).astype(np.float64)
    m = np.hypot(x.astype(np.float64), y.astype(np.float64))
    got_c, got_s = a * x / m, a * y / m
    # against the routine in float32 as it runs: 5e-7, which is the routine's own rounding of the angle (a float32 of half a
    # turn resolves 6e-8 turns = 3.7e-7 rad) ...
    assert np.abs(got_c - want_c).max() < 6e-7 and np.abs(got_s - want_s).max() < 6e-7
    # ... and against the same interpolation carried out in float64, where only the formula is left: 1e-7
    t64 = np.arctan2(y.astype(np.float64), x.astype(np.float64)) / (2 * np.pi)

    def table64(t):
        fi = 512.0 * (t - np.floor(t))
        idx = np.minimum(fi.astype(np.int64), 511)
        return (1.0 - (fi - idx)) * tab[idx] + (fi - idx) * tab[idx + 1]
    assert np.abs(got_c - table64(t64 + 0.25)).max() < 1e-7 and np.abs(got_s - table64(t64)).max() < 1e-7
    assert np.abs(x / m - want_c).max() > 1.5e-5                                 # what separates the stage's two forms


def test_engine_kernels_division_free_forms_are_the_quotients():
    """csrc/rdsp_engine.hip replaces two IEEE double divisions of the engine's arithmetic by forms without one and claims
    the same bits: v / 32767.0 for every int16 v (q0 = v y, r = fma(-q0, 32767, v), q = fma(r, y, q0), y = RN(1 / 32767)),
    and trunc(a / d) for the table index (d = the double of the float 2 pi; the estimate a (1 / d) corrected by two exact
    comparisons with k d).  Both are checked here in exact rational arithmetic: all 65 536 int16 values; every index k with
    a = k d and its two neighbouring doubles, and two million random phases."""
    from fractions import Fraction
    y = 1.0 / 32767.0

    def fma(a, b, c):
        return float(Fraction(a) * Fraction(b) + Fraction(c))      # exact, then one rounding
    for v in range(-32768, 32768):
        q0 = float(v) * y
        assert fma(fma(-q0, 32767.0, float(v)), y, q0) == v / 32767.0, v
    d = float(np.float32(6.2831854820251465))

    def index(a):
        k = np.trunc(a * (1.0 / d)).astype(np.int64)
        k = np.where(k.astype(np.float64) * d > a, k - 1, k)
        return np.where((k + 1).astype(np.float64) * d <= a, k + 1, k)
    n = np.arange(0, 65536, dtype=np.float64)
    assert all(Fraction(float(k) * d) == Fraction(int(k)) * Fraction(d) for k in n[::97])   # k d is exact in double
    for a in (n * d, np.nextafter(n * d, np.inf), np.nextafter(n[1:] * d, -np.inf)):
        assert np.array_equal(index(a), np.trunc(a / d).astype(np.int64))
    ph = np.random.default_rng(3).uniform(0, 6.2831855, 2_000_000).astype(np.float32).astype(np.float64)
    a = ph * 65535.0
    assert np.array_equal(index(a), np.trunc(a / d).astype(np.int64))
