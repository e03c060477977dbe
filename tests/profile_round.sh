#!/bin/bash
# One GPU call: rocprofv3 kernel-trace stats of the bench lines and the PMC passes of every
# configuration (FETCH_SIZE and WRITE_SIZE in separate passes, SQ counters in a third), all under
# gpurun_out/prof_<cfg>/ and gpurun_out/pmc_<cfg>_<pass>/; tests/summarize_profiles.py turns them
# into profiles/<round>_* and profiles/counters.json.
# usage: bash tests/profile_round.sh [configs...]   (on the GPU box)
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
CFGS=${@:-K3 K2 K4 K5 F1 ENGINE}
B="--steps 100 --warmup 20 --no-cpu-baseline --no-host-io --no-iso --no-extra-legs"
P="--steps 5 --warmup 2 --no-cpu-baseline --no-host-io --no-kernel-timing --no-extra-legs"
for K in $CFGS; do
  python3 -c "import bench; print(bench.lib_sha())" > $OUT/prof_$K.sha   # the sources these counters belong to
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$K -o trace -- python3 $ROOT/bench.py --config $K $B > $OUT/prof_$K.json 2> $OUT/prof_$K.err) || { echo "rocprof $K failed"; tail -5 $OUT/prof_$K.err; }
  echo "trace $K done"
done
# the optional stages of row F3 at the K3 shape: SAM groups (tests/micro/f3_bench.py) and the IIR filter bank
if [[ " $CFGS " == *" K3 "* ]]; then
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_F3SAM -o trace -- python3 $ROOT/tests/micro/f3_bench.py > $OUT/prof_F3SAM.log 2> $OUT/prof_F3SAM.err) || { echo "rocprof F3SAM failed"; tail -5 $OUT/prof_F3SAM.err; }
  (cd /tmp && RDSP_AUDIO_IIR=2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_K3IIR -o trace -- python3 $ROOT/bench.py --config K3 $B > $OUT/prof_K3IIR.json 2> $OUT/prof_K3IIR.err) || { echo "rocprof K3IIR failed"; tail -5 $OUT/prof_K3IIR.err; }
  echo "trace F3 stages done"
fi
# PMC: one counter family per pass (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2), kernel trace only
for K in $CFGS; do
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_${K}_$C -o pmc -- python3 $ROOT/bench.py --config $K $P > $OUT/pmc_${K}_$C.json 2> $OUT/pmc_${K}_$C.err) || { echo "pmc $K $C failed"; tail -5 $OUT/pmc_${K}_$C.err; }
  done
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/pmc_${K}_SQ -o pmc -- python3 $ROOT/bench.py --config $K $P > $OUT/pmc_${K}_SQ.json 2> $OUT/pmc_${K}_SQ.err) || { echo "pmc $K SQ failed"; tail -5 $OUT/pmc_${K}_SQ.err; }
  echo "pmc $K done"
done
