for fv in 1 0; do for pr in 2,1 2,0 1,1 3,1; do
RDSP_FRONT_VARIANT=$fv RDSP_PRIO=$pr python bench.py --config K3 --steps 10 --warmup 2 --no-cpu-baseline --no-host-io > gpurun_out/fv.json 2>gpurun_out/fv.err || tail -3 gpurun_out/fv.err
python - "$fv" "$pr" <<PY
import json,sys
d=json.loads(open("gpurun_out/fv.json").read().strip().splitlines()[-1])
print("front lean", sys.argv[1], "prio", sys.argv[2], "ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done
for fv in 1 0; do RDSP_FRONT_VARIANT=$fv python bench.py --config K5 --steps 10 --warmup 2 --no-cpu-baseline --no-host-io 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K5 lean', $fv, round(d['ms_per_step'],3))"; done
