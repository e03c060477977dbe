#!/bin/bash
# what the tail chain idles on between two steps of pipelined K3: kernel timeline with and without the
# per-kernel HIP events in the timed region (tests/micro/gap_trace.py), and the bench line both ways
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
B="--config ${CFG:-K3} --steps 100 --warmup 20 --no-cpu-baseline --no-host-io --no-iso --no-extra-legs"
for fl in "" "--no-kernel-timing"; do
  tag=gap$( [ -z "$fl" ] && echo _ev || echo _noev )
  python3 bench.py $B $fl 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench [$fl] ms/step %.4f kernels %s'%(d['ms_per_step'], d['kernels_ms_per_step']))"
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o t -- python3 $ROOT/bench.py $B $fl > $OUT/$tag.json 2> $OUT/$tag.err) || { echo "rocprof failed"; tail -3 $OUT/$tag.err; }
  f=$(find $OUT/$tag -name '*kernel_trace.csv' | head -1)
  echo "[$fl] $(python3 tests/micro/gap_trace.py $f)"
  rm -rf $OUT/$tag
done
