for a in "--groups 1" "--groups 16" "--groups 256" "--groups 16 --retune-every 1" "--groups 256 --retune-every 1"; do
python bench.py --config K3 --steps 20 --warmup 3 --no-cpu-baseline $a > gpurun_out/g.json 2> gpurun_out/g.err || { tail -5 gpurun_out/g.err; exit 1; }
python - "$a" <<PY
import json,sys
d=json.loads(open("gpurun_out/g.json").read().strip().splitlines()[-1])
print(sys.argv[1], "| %.1f Msamples/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernels_ms_per_step"].items()})
PY
done
