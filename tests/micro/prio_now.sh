# pipelined K3, product kernels: tail-kernel wave priority 0..3 (the frequency-domain front kernel never raises its own)
for rep in 1 2; do for pr in 0,0 0,1 0,2 0,3; do
RDSP_PRIO=$pr python bench.py --config K3 --steps 40 --warmup 8 --no-cpu-baseline --no-host-io > gpurun_out/pr.json 2>gpurun_out/pr.err || tail -3 gpurun_out/pr.err
python - "$pr" <<PY
import json,sys
d=json.loads(open("gpurun_out/pr.json").read().strip().splitlines()[-1])
print("prio(front FIR, tail)", sys.argv[1], "ms/step %.4f"%d["ms_per_step"], {k:round(v,4) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done
