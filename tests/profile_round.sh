#!/bin/bash
# One GPU call: rocprofv3 kernel-trace stats of the bench lines and the PMC passes of the K3
# workload (FETCH_SIZE and WRITE_SIZE in separate passes, SQ counters in a third), all under
# gpurun_out/prof_<tag>/; tests/summarize_profiles.py turns them into profiles/<round>_*.
# usage: bash tests/profile_round.sh   (on the GPU box; `cd /tmp && export TMPDIR=/tmp` per the guide)
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
B="--steps 100 --warmup 20 --no-cpu-baseline --no-host-io --no-iso"
for K in K3 K2 K4 F1 K5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$K -o trace -- python3 $ROOT/bench.py --config $K $B > $OUT/prof_$K.json 2> $OUT/prof_$K.err) || { echo "rocprof $K failed"; tail -5 $OUT/prof_$K.err; }
  echo "trace $K done"
done
# PMC: one counter family per pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), no tracing domains besides the kernel trace
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$C -o pmc -- python3 $ROOT/bench.py --config K3 --steps 5 --warmup 2 --no-cpu-baseline --no-host-io --no-kernel-timing > $OUT/pmc_$C.json 2> $OUT/pmc_$C.err) || { echo "pmc $C failed"; tail -5 $OUT/pmc_$C.err; }
  echo "pmc $C done"
done
(cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_SQ -o pmc -- python3 $ROOT/bench.py --config K3 --steps 5 --warmup 2 --no-cpu-baseline --no-host-io --no-kernel-timing > $OUT/pmc_SQ.json 2> $OUT/pmc_SQ.err) || { echo "pmc SQ failed"; tail -5 $OUT/pmc_SQ.err; }
echo "pmc SQ done"
find $OUT -name "*.csv" | head -40
