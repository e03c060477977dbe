#!/bin/bash
# cost of the per-launch HIP events inside the timed region (pipelined K3)
for rep in 1 2; do for fl in "" "--no-kernel-timing"; do
RDSP_FRONT_VARIANT=${FV:-0} RDSP_PRIO=${PR:-2,2} python bench.py --config K3 --steps 20 --warmup 3 --no-cpu-baseline --no-host-io $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [$fl] ms/step %.3f'%d['ms_per_step'])"
done; done
