#!/bin/bash
# same-box A/B: K3 in the benchmark form (448-sample frames), fewer front workgroups per CU by LDS padding
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for pad in 0 4096 8192 12288; do
  RDSP_FD_LDS_PAD=$pad timeout -k 10 120 python bench.py --no-extra-legs --no-cpu-baseline --no-host-io --steps 60 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('pad $pad rep $rep', round(d['ms_per_step'],4), round(d['ms_per_step_steady'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items()})" >> gpurun_out/r6_pad_sweep7.log
done
done
cat gpurun_out/r6_pad_sweep7.log
