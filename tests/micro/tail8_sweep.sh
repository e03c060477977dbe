#!/bin/bash
# pipelined K3: half-row tail kernel (8 lanes per channel) against the default, over priorities
for tv in 16r 8r; do for pr in ${PRIOS:-2,2 2,3 1,2 0,3 2,1}; do
RDSP_TAIL_VARIANT=$tv RDSP_PRIO=$pr python bench.py --config ${K:-K3} --steps ${STEPS:-100} --warmup ${WARMUP:-20} --no-cpu-baseline --no-host-io 2>gpurun_out/t8.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail $tv prio $pr ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()}, {k:round(v,3) for k,v in (d['kernels_ms_isolated'] or {}).items()})" || tail -3 gpurun_out/t8.err
done; done
