#!/bin/bash
# does the step time depend on how long the GPU has been busy before the timed region?
for rep in 1 2; do for wk in "3 20" "30 20" "100 20" "3 200" "100 200"; do set -- $wk
RDSP_FRONT_VARIANT=${FV:-0} RDSP_PRIO=${PR:-2,2} python bench.py --config K3 --warmup $1 --steps $2 --no-cpu-baseline --no-host-io 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup $1 steps $2 ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
done; done
