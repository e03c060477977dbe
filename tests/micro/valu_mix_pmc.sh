#!/bin/bash
# Executed VALU instruction mix of the bench kernels (PMC): bash tests/micro/valu_mix_pmc.sh [configs...]  (default K3)
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3/valumix; mkdir -p $OUT; export TMPDIR=/tmp
P="--steps 4 --warmup 2 --no-cpu-baseline --no-host-io --no-kernel-timing --no-iso --no-pipeline"
for K in ${@:-K3}; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU -d $OUT/a_$K -o pmc -- python3 $ROOT/bench.py --config $K $P > $OUT/a_$K.log 2>&1) || echo "pass a $K failed"
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.getcwd(), "gpurun_out/r3/valumix")
for d in sorted(glob.glob(out + "/a_*")):
    if not os.path.isdir(d): continue
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"]
        if "front" not in k and "tail" not in k and "spectrum" not in k: continue
        k = "front" if "front" in k else ("tail" if "tail" in k else "spectrum")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k in acc:
        c = {cn: v / len(n[k]) for cn, v in acc[k].items()}
        t = c.get("SQ_INSTS_VALU", 1)
        print(os.path.basename(d), k, " ".join("%s=%.4g(%.3f)" % (cn.replace("SQ_INSTS_", ""), v, v / t) for cn, v in sorted(c.items())))
PY
