#!/bin/bash
# pipelined configs: lean vs full front kernel at given priorities
for K in ${CONFIGS:-K3 K5 K4}; do for fv in 1 0; do for pr in ${PRIOS:-2,2 2,1}; do
RDSP_FRONT_VARIANT=$fv RDSP_PRIO=$pr python bench.py --config $K --steps ${STEPS:-100} --warmup ${WARMUP:-20} --no-cpu-baseline --no-host-io 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$K lean $fv prio $pr ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
done; done; done
