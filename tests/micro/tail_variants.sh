python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "nlms or split or isolated" 2>&1 | tail -4
for v in 16 16r; do for K in K3 K5; do for m in "" "--no-pipeline"; do RDSP_TAIL_VARIANT=$v python bench.py --config $K --steps 10 --warmup 2 --no-cpu-baseline --no-host-io $m > gpurun_out/tm.json 2>gpurun_out/tm.err || tail -3 gpurun_out/tm.err
python - "$v" "$K" "$m" <<PY
import json,sys
d=json.loads(open("gpurun_out/tm.json").read().strip().splitlines()[-1])
print("tail", sys.argv[1], sys.argv[2], sys.argv[3] or "pipelined", "ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done; done
