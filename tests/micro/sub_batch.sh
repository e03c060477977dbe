#!/bin/bash
# K5 (8192 channels): pipelined step with and without channel sub-batches (RDSP_SUB_BATCH=0 turns them off)
for rep in 1 2; do for sb in 0 4096 2048; do
RDSP_SUB_BATCH=$sb python bench.py --config K5 --steps ${STEPS:-100} --warmup ${WARMUP:-20} --no-cpu-baseline --no-host-io --no-iso 2>gpurun_out/sb.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sub_batch $sb ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})" || tail -3 gpurun_out/sb.err
done; done
