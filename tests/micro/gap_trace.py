#!/usr/bin/env python3
"""Timeline of pipelined K3 from a rocprofv3 --kernel-trace CSV: per step, when the tail kernel of
call k ends, when the front kernel of call k+1 ends and when the tail kernel of call k+1 starts --
i.e. what the tail chain (the floor under the step) idles on.  usage: gap_trace.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
tails = sorted((s, e) for n, s, e in ks if "rdsp_tail" in n)
fronts = sorted((s, e) for n, s, e in ks if "rdsp_front" in n)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tails, fronts = tails[skip:], fronts[skip:]
n = min(len(tails), len(fronts)) - 1
gap_t, gap_f, lead, dur_t, dur_f, per = [], [], [], [], [], []
for i in range(1, n):
    gap_t.append(tails[i][0] - tails[i - 1][1])       # tail chain idle between consecutive tails
    gap_f.append(fronts[i][0] - fronts[i - 1][1])     # front chain idle
    dur_t.append(tails[i][1] - tails[i][0])
    dur_f.append(fronts[i][1] - fronts[i][0])
    per.append(tails[i][1] - tails[i - 1][1])
    # front kernel whose end is the last one before this tail's start: how long before?
    fe = [e for s, e in fronts if e <= tails[i][0]]
    lead.append(tails[i][0] - max(fe) if fe else -1)
m = lambda v: sum(v) / max(len(v), 1) / 1e3
print(f"steps {n - 1}: period {m(per):.1f} us  tail {m(dur_t):.1f}  front {m(dur_f):.1f}  tail-chain idle {m(gap_t):.1f}  "
      f"front-chain idle {m(gap_f):.1f}  newest finished front ended {m(lead):.1f} us before the tail started")
