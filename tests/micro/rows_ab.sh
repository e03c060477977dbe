#!/bin/bash
# same-box A/B: decimator forms (RDSP_FIR_VARIANT: 2 = 448-sample frames, -1 = one granule per frame, 5 / 6 = rows)
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/rows_ab.log
for rep in 1 2; do
for cfg in ${CFGS:-K2 K4 K3}; do
for v in ${VARIANTS:-2 -1 5 6}; do
  RDSP_FIR_VARIANT=$v timeout -k 10 120 python bench.py --config $cfg --no-extra-legs --no-cpu-baseline --no-host-io --steps 60 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$cfg variant $v rep $rep', round(d['ms_per_step'],4), round(d.get('ms_per_step_steady',0),4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items()})" >> $LOG
done
done
done
cat $LOG
