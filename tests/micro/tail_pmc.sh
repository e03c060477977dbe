#!/bin/bash
# PMC passes of the tail kernel alone (tests/micro/tail_bench): the product (variant 100); with an EXPERIMENTAL=1
# build of the library also 104 (weights one block stale) and 105 (four steps per reduction).
# usage (GPU box, repository root): bash tests/micro/tail_pmc.sh [variants...]
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/tailpmc; mkdir -p $OUT; export TMPDIR=/tmp
for V in ${@:-100}; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/a_$V -o pmc -- $ROOT/tests/micro/tail_bench 4096 $V > $OUT/a_$V.log 2>&1) || echo "pass a $V failed"
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU -d $OUT/b_$V -o pmc -- $ROOT/tests/micro/tail_bench 4096 $V > $OUT/b_$V.log 2>&1) || echo "pass b $V failed"
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.getcwd(), "gpurun_out/r3/tailpmc")
for d in sorted(glob.glob(out + "/*_1??")):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"][:40]; acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k in acc: print(os.path.basename(d), k, {c: round(v / len(n[k])) for c, v in acc[k].items()})
PY
