#!/bin/bash
# Why the library's default decimator (one granule per frame) shows 1.34 - 1.83 ms per K3 step where the 448-sample form
# holds 1.14 +- 0.02: per-launch kernel durations over a long run (rocprofv3 kernel trace), tail priority 2 and 0.
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
for pr in 2,2 2,0; do
  (cd /tmp && RDSP_FIR_VARIANT=-1 RDSP_PRIO=$pr rocprofv3 --kernel-trace --output-format csv -d $OUT/dft -o t -- python3 $ROOT/bench.py --config K3 --steps 300 --warmup 10 --no-cpu-baseline --no-host-io --no-iso --no-extra-legs --no-kernel-timing > $OUT/dft.json 2> $OUT/dft.err) || { echo "rocprof failed"; tail -3 $OUT/dft.err; }
  f=$(find $OUT/dft -name '*kernel_trace.csv' | head -1)
  echo "tail prio $pr: $(python3 -c "import json;d=json.loads(open('$OUT/dft.json').read().strip().splitlines()[-1]);print('ms/step',round(d['ms_per_step'],3))")"
  python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for key in ("rdsp_tail", "rdsp_front"):
    k = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"])
    d = [(e - s) / 1e3 for s, e in k]
    per = [(k[i][1] - k[i - 1][1]) / 1e3 for i in range(1, len(k))]
    print(key, len(k), "launches; duration us, every 10th:", " ".join(f"{x:.0f}" for x in d[::10]))
    print(key, "period between ends us, every 10th:", " ".join(f"{x:.0f}" for x in per[::10]))
PY
  rm -rf $OUT/dft
done
