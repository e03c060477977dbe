# K3 step time against the call size (blocks per call): the fixed cost per call is the intercept
for rep in 1 2; do for B in 128 256 512 1024; do
python bench.py --config K3 --blocks $B --steps 40 --warmup 8 --no-cpu-baseline --no-host-io > gpurun_out/cs.json 2>gpurun_out/cs.err || tail -3 gpurun_out/cs.err
python - "$B" <<PY
import json,sys
d=json.loads(open("gpurun_out/cs.json").read().strip().splitlines()[-1]); B=int(sys.argv[1])
print("blocks", B, "ms/step %.4f"%d["ms_per_step"], "per 512 blocks %.4f"%(d["ms_per_step"]*512/B), {k:round(v,4) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done
