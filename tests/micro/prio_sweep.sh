# pipelined K3: sweep the wave priorities for both tail kernels
for v in 16 16r; do for pr in 2,0 2,1 1,0 1,1 0,0 0,1 3,1 3,2 2,2; do
RDSP_TAIL_VARIANT=$v RDSP_PRIO=$pr python bench.py --config K3 --steps 10 --warmup 2 --no-cpu-baseline --no-host-io > gpurun_out/pr.json 2>gpurun_out/pr.err || tail -3 gpurun_out/pr.err
python - "$v" "$pr" <<PY
import json,sys
d=json.loads(open("gpurun_out/pr.json").read().strip().splitlines()[-1])
print("tail", sys.argv[1], "prio(front FIR, tail)", sys.argv[2], "ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done
