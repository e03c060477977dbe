#!/bin/bash
# Same-box A/B of built libraries: bash tests/micro/lib_ab.sh variants/a.so variants/b.so ...
# (each library is passed to bench.py as --lib; configs via CONFIGS="K3 K2", extra bench flags via BFLAGS)
mkdir -p gpurun_out
for rep in 1 2; do for lib in "$@"; do for K in ${CONFIGS:-K3}; do
  python bench.py --lib $PWD/$lib --config $K --steps ${STEPS:-10} --warmup ${WARMUP:-2} --no-cpu-baseline --no-host-io --no-extra-legs $BFLAGS > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python - "$lib" "$K" <<PY
import json,sys
d=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "ms/step %.3f"%d["ms_per_step"], {k:round(v,3) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done; done
