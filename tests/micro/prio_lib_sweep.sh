#!/bin/bash
# pipelined K3, one library (LIB=path, passed as --lib), sweep front variant x priorities
for fv in ${FVS:-1 0}; do for pr in ${PRIOS:-2,1 1,1 0,1 2,2 1,2 0,0 3,1 3,2}; do
RDSP_FRONT_VARIANT=$fv RDSP_PRIO=$pr python bench.py ${LIB:+--lib $LIB} --config K3 --steps ${STEPS:-10} --warmup ${WARMUP:-2} --no-cpu-baseline --no-host-io --no-iso > gpurun_out/pr.json 2>gpurun_out/pr.err || tail -3 gpurun_out/pr.err
python - "$fv" "$pr" <<PY
import json,sys
d=json.loads(open("gpurun_out/pr.json").read().strip().splitlines()[-1])
print("lean", sys.argv[1], "prio(front FIR, tail)", sys.argv[2], "ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done
