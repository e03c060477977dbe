#!/bin/bash
# K5 (8192 channels): half-row tail (one wave per SIMD at this size) against sub-batches of the 16-lane one
run() { env "$@" python bench.py --config K5 --steps ${STEPS:-100} --warmup 20 --no-cpu-baseline --no-host-io --no-iso 2>gpurun_out/k5h.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})" || tail -3 gpurun_out/k5h.err; }
run RDSP_SUB_BATCH=4096
for fv in 1 0; do for pr in 2,2 2,3 1,2; do
run RDSP_SUB_BATCH=0 RDSP_TAIL_VARIANT=8r RDSP_FRONT_VARIANT=$fv RDSP_PRIO=$pr
done; done
run RDSP_SUB_BATCH=4096
