#!/bin/bash
# As lib_ab.sh with REPS interleaved repetitions and the mean per library at the end:
#   REPS=4 CONFIGS="K3" bash tests/micro/lib_ab3.sh variants/a.so variants/b.so
mkdir -p gpurun_out; : > gpurun_out/ab3.txt
for rep in $(seq 1 ${REPS:-3}); do for lib in "$@"; do for K in ${CONFIGS:-K3}; do
  python bench.py --lib $PWD/$lib --config $K --steps ${STEPS:-50} --warmup ${WARMUP:-10} --no-cpu-baseline --no-host-io --no-extra-legs $BFLAGS > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python - "$lib" "$K" <<PY | tee -a gpurun_out/ab3.txt
import json,sys
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
k=d["kernels_ms_per_step"]
print(sys.argv[1], sys.argv[2], "%.4f"%d["ms_per_step"], "%.4f"%k.get("rdsp_front_fd_kernel",0), "%.4f"%k.get("rdsp_tail_kernel",0))
PY
done; done; done
python - <<PY
import collections
acc=collections.defaultdict(list)
for l in open("gpurun_out/ab3.txt"):
    f=l.split(); acc[(f[0],f[1])].append([float(x) for x in f[2:]])
for k,v in acc.items():
    n=len(v); print("MEAN", k[0], k[1], "step %.4f front %.4f tail %.4f"%tuple(sum(r[i] for r in v)/n for i in range(3)), "min step %.4f"%min(r[0] for r in v), "n=%d"%n)
PY
