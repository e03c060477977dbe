#!/bin/bash
# PMC of the front kernel at K2 / K4 under the decimator forms (RDSP_FIR_VARIANT)
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
P="--steps 5 --warmup 2 --no-cpu-baseline --no-host-io --no-kernel-timing --no-extra-legs"
for cfg in ${CFGS:-K2}; do
for v in ${VARIANTS:-2 5 6}; do
  export RDSP_FIR_VARIANT=$v
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/rows_pmc_${cfg}_$v -o pmc -- python3 $ROOT/bench.py --config $cfg $P > $OUT/rows_pmc_${cfg}_$v.json 2> $OUT/rows_pmc_${cfg}_$v.err) || { echo "pmc $cfg $v failed"; tail -5 $OUT/rows_pmc_${cfg}_$v.err; }
  F=$(find $OUT/rows_pmc_${cfg}_$v -name "*counter_collection.csv" | head -1)
  test -n "$F" && python3 - "$F" $cfg $v <<'PY'
import csv,sys,collections
f,cfg,v=sys.argv[1:4]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name']
    if 'front' not in k: continue
    acc[k[:40]][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_INSTS_VALU': n[k[:40]]+=1
for k,d in acc.items():
    print(cfg, 'variant', v, k, 'launches', n[k], {c: '%.3e'%(x/n[k]) for c,x in d.items()})
PY
done
done
