#!/bin/bash
# same-box A/B at K2: the four-frame filter stage (226-256 VGPRs, two waves per SIMD) against the one-frame form,
# full (177 VGPRs: two waves) and lean (165: three waves per SIMD, 12 x 12.9 KiB of LDS per CU)
cd $GRAFT_REPO_ROOT
LOG=gpurun_out/k2_occupancy.log
for rep in 1 2; do
for sw in "RDSP_NO_QUAD=0 RDSP_FRONT_VARIANT=0" "RDSP_NO_QUAD=0 RDSP_FRONT_VARIANT=1" "RDSP_NO_QUAD=1 RDSP_FRONT_VARIANT=0" "RDSP_NO_QUAD=1 RDSP_FRONT_VARIANT=1"; do
  env $sw timeout -k 10 120 python bench.py --config K2 --no-extra-legs --no-cpu-baseline --no-host-io --steps 60 --warmup 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('K2 $sw rep $rep', round(d['ms_per_step'],4), {k:round(v,4) for k,v in d['kernels_ms_per_step'].items()})" >> $LOG
done
done
cat $LOG
