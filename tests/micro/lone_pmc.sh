#!/bin/bash
# what SQ_WAIT_ANY / SQ_WAIT_INST_ANY count: PMC pass over the modes of tests/micro/lone_wave
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/lonepmc; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/a -o pmc -- $ROOT/tests/micro/lone_wave > $OUT/a.log 2>&1) || echo "pass failed"
python3 - <<'PY'
import csv, glob, collections, os
fs = glob.glob(os.getcwd() + "/gpurun_out/r3/lonepmc/a/**/*counter_collection.csv", recursive=True)
rows = collections.OrderedDict()
for row in csv.DictReader(open(fs[0])):
    key = (row["Kernel_Name"], row["Dispatch_Id"]); rows.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
seen = collections.Counter()
for (k, d), c in rows.items():
    seen[k] += 1
    if seen[k] != 2: continue   # second dispatch of a mode = the timed one at 1 wave per SIMD
    w = c.get("SQ_WAVE_CYCLES", 1)
    print(k[-28:], "wave_quads %.3g  active_any %.2f  wait_any %.2f  wait_inst %.2f  valu/quad %.2f" % (w, c.get("SQ_ACTIVE_INST_ANY",0)/w, c.get("SQ_WAIT_ANY",0)/w, c.get("SQ_WAIT_INST_ANY",0)/w, c.get("SQ_INSTS_VALU",0)/w))
PY
