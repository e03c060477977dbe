#!/bin/bash
# PMC passes of the front kernel in the bench configurations (instruction mix, wait breakdown, LDS):
# usage (GPU box, repository root): bash tests/micro/front_pmc.sh [configs...]     (default K2 K3)
# LIB=path selects a library (passed to bench.py as --lib; default: the in-tree build); results under gpurun_out/r3/frontpmc/, one summary line per kernel and pass.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/frontpmc; mkdir -p $OUT; export TMPDIR=/tmp
P="--steps 4 --warmup 2 --no-cpu-baseline --no-host-io --no-kernel-timing --no-iso --no-pipeline"
for K in ${@:-K2 K3}; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_FLAT -d $OUT/a_$K -o pmc -- python3 $ROOT/bench.py ${LIB:+--lib $LIB} --config $K $P > $OUT/a_$K.log 2>&1) || echo "pass a $K failed"
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $OUT/b_$K -o pmc -- python3 $ROOT/bench.py ${LIB:+--lib $LIB} --config $K $P > $OUT/b_$K.log 2>&1) || echo "pass b $K failed"
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH -d $OUT/c_$K -o pmc -- python3 $ROOT/bench.py ${LIB:+--lib $LIB} --config $K $P > $OUT/c_$K.log 2>&1) || echo "pass c $K failed"
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.getcwd(), "gpurun_out/r3/frontpmc")
for d in sorted(glob.glob(out + "/[abc]_*")):
    if not os.path.isdir(d): continue
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"]
        if "front" not in k and "tail" not in k: continue
        k = "front" if "front" in k else "tail"
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k in acc:
        c = {cn: v / len(n[k]) for cn, v in acc[k].items()}
        w = c.get("SQ_WAVE_CYCLES", 0) or 1
        print(os.path.basename(d), k, " ".join("%s=%.4g(%.3f/wq)" % (cn.replace("SQ_", ""), v, v / w) for cn, v in sorted(c.items())))
PY
