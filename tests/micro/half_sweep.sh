#!/bin/bash
# pipelined K3: tail layout x front variant x priorities
for tv in ${TVS:-16r 8r}; do for fv in 0 1; do for pr in ${PRIOS:-2,2 2,3}; do
RDSP_TAIL_VARIANT=$tv RDSP_FRONT_VARIANT=$fv RDSP_PRIO=$pr python bench.py --config ${K:-K3} --steps ${STEPS:-100} --warmup ${WARMUP:-20} --no-cpu-baseline --no-host-io --no-iso 2>gpurun_out/t8.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail $tv lean $fv prio $pr ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})" || tail -3 gpurun_out/t8.err
done; done; done
