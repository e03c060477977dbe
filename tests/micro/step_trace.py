#!/usr/bin/env python3
"""Per-launch durations of the tail and front kernels in launch order, from a rocprofv3 --kernel-trace CSV:
what the first steps of a timed region cost against the settled ones.  usage: step_trace.py <kernel_trace.csv> [last N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for key in ("rdsp_tail", "rdsp_front"):
    k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"])
    d = [(e - s) / 1e3 for s, e in k][-n:]
    print(key, len(k), "launches; last", len(d), "durations (us):", " ".join(f"{x:.0f}" for x in d))
