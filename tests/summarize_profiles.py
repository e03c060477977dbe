"""Turn the rocprofv3 output of tests/profile_round.sh (gpurun_out/prof_*, pmc_*) into the
tracked summaries under profiles/: <round>_<cfg>_kernel_stats.csv, <round>_<cfg>_bench.json,
<round>_<cfg>_pmc_summary.txt and counters.json (per configuration and kernel: HBM bytes per
launch -- FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, WRITE_SIZE as read --
VALU instructions and the VALU-busy fraction), which bench.py reads for roofline.traffic.
usage: python tests/summarize_profiles.py r02"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def short(name):
    if "rdsp_tail" in name and "engine" not in name:
        return "rdsp_tail_kernel"
    for k in ("rdsp_engine_front_pipe_kernel", "rdsp_engine_tail_pipe_kernel", "rdsp_engine_hilbert_kernel", "rdsp_engine_front_kernel", "rdsp_engine_tail_kernel"):
        if k in name:
            return k
    for k in ("rdsp_front_fd_kernel", "rdsp_front_kernel", "rdsp_tail_kernel", "rdsp_sam_kernel", "rdsp_spectrum_kernel", "rdsp_group_store_kernel"):
        if k in name:
            return k
    return None


def pmc_per_launch(tag):
    """mean counter value per dispatch, per kernel"""
    f = newest(os.path.join(OUT, f"pmc_{tag}", "**", "*counter_collection.csv"))
    if not f:
        return {}
    acc = defaultdict(lambda: defaultdict(list))
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row.get("Kernel_Name", ""))
            if k:
                acc[k][row["Counter_Name"]].append((row.get("Dispatch_Id"), float(row["Counter_Value"])))
    res = {}
    for k, counters in acc.items():
        res[k] = {}
        for cname, vals in counters.items():
            per_dispatch = defaultdict(float)
            for d, v in vals:
                per_dispatch[d] += v          # rows are per XCD / instance: sum them
            res[k][cname] = sum(per_dispatch.values()) / len(per_dispatch)
    return res


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
    os.makedirs(PROF, exist_ok=True)
    for cfg in ("K3", "K2", "K4", "F1", "K5", "ENGINE"):
        stats = newest(os.path.join(OUT, f"prof_{cfg}", "**", "*kernel_stats.csv"))
        if stats:
            shutil.copy(stats, os.path.join(PROF, f"{rnd}_{cfg.lower()}_kernel_stats.csv"))
        bj = os.path.join(OUT, f"prof_{cfg}.json")
        if os.path.exists(bj):
            lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
            if lines:
                open(os.path.join(PROF, f"{rnd}_{cfg.lower()}_bench.json"), "w").write(lines[-1] + "\n")
    for tag, name in (("F3SAM", "f3_sam"), ("K3IIR", "k3_iir")):   # optional stages of row F3 at the K3 shape
        stats = newest(os.path.join(OUT, f"prof_{tag}", "**", "*kernel_stats.csv"))
        if stats:
            shutil.copy(stats, os.path.join(PROF, f"{rnd}_{name}_kernel_stats.csv"))
        for ext in ("log", "json"):
            f = os.path.join(OUT, f"prof_{tag}.{ext}")
            if os.path.exists(f):
                lines = [l for l in open(f).read().splitlines() if l.startswith("{") or l.startswith("K3 ") or l.startswith("K4 ")]
                if lines:
                    open(os.path.join(PROF, f"{rnd}_{name}_bench.{'json' if ext == 'json' else 'txt'}"), "w").write("\n".join(lines[-8:] if ext == "log" else lines[-1:]) + "\n")
    cpath = os.path.join(PROF, "counters.json")
    counters = json.load(open(cpath)) if os.path.exists(cpath) else {}
    for cfg in ("K2", "K3", "K4", "K5", "F1", "ENGINE"):
        fetch, write, sq = (pmc_per_launch(f"{cfg}_{t}") for t in ("FETCH_SIZE", "WRITE_SIZE", "SQ"))
        if not (fetch or write or sq):
            continue
        counters[cfg] = {"round": rnd}
        shaf = os.path.join(OUT, f"prof_{cfg}.sha")   # hash of the library sources the passes ran on (bench.lib_sha)
        if os.path.exists(shaf):
            counters[cfg]["lib_sha"] = open(shaf).read().strip()
        lines = [f"{cfg} per-launch PMC (rocprofv3, separate passes; FETCH_SIZE doubled per the gfx950 note in "
                 "MI355X_MICROARCH.md, WRITE_SIZE as read; units of both: KB):"]
        for k in sorted(set(fetch) | set(write) | set(sq)):
            rd = 2.0 * fetch.get(k, {}).get("FETCH_SIZE", 0.0) * 1024.0
            wr = write.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
            c = sq.get(k, {})
            ent = {"hbm_bytes": rd + wr, "hbm_read_bytes": rd, "hbm_write_bytes": wr}
            if c.get("GRBM_GUI_ACTIVE"):
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_ACTIVE_INST_VALU counts quad-cycles over
                # all 1024 SIMDs: busy fraction = 4 * active / (1024 SIMDs * cycles of the launch)
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0
                ent.update(valu_insts=c.get("SQ_INSTS_VALU"), lds_insts=c.get("SQ_INSTS_LDS"), gpu_cycles=cyc,
                           valu_busy_frac=4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * cyc))
            counters[cfg][k] = ent
            lines.append(f"  {k}: read {rd / 1e9:.3f} GB (FETCH_SIZE {fetch.get(k, {}).get('FETCH_SIZE', 0):.0f} KB x2), "
                         f"write {wr / 1e9:.3f} GB, total {(rd + wr) / 1e9:.3f} GB"
                         + (f", VALU busy {ent['valu_busy_frac']:.3f}" if "valu_busy_frac" in ent else ""))
        for k, c in sorted(sq.items()):
            lines.append(f"  {k}: " + ", ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
        open(os.path.join(PROF, f"{rnd}_{cfg.lower()}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))
    json.dump(counters, open(cpath, "w"), indent=1)


if __name__ == "__main__":
    main()
