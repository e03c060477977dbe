#!/bin/bash
# One GPU call: parity tests, then K3/K2/K4 bench lines (gpurun_out/).
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
python bench.py --config K3 --steps 100 --warmup 20 --no-cpu-baseline --no-pipeline > gpurun_out/bench_K3np.json 2> gpurun_out/bench_K3np.err
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_K3np.json").read().strip().splitlines()[-1])
print("K3 no-pipeline", "%.1f Msamples/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], d["kernels_ms_per_step"])
PY
for K in K3 K2 K4; do
  python bench.py --config $K --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_$K.json 2> gpurun_out/bench_$K.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_$K.json").read().strip().splitlines()[-1])
print("$K", "%.1f Msamples/s"%d["value"], "ms/step %.3f"%d["ms_per_step"], d["kernels_ms_per_step"], "roofline frac %.4f"%d["roofline"]["frac"], "chain frac %.4f"%d["chain_hbm"]["frac_of_peak"])
PY
done
