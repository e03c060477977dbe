# pipelined K3 / K5: tail-kernel wave priority and tail-stream queue priority A/B, interleaved repetitions
for rep in 1 2 3; do for cfgv in "K3 -1" "K3 2" "K5 2"; do set -- $cfgv; for pr in "2,2 -" "2,0 -" "2,2 1" "2,0 1" "2,2 0"; do set -- $1 $2 $pr
if [ "$4" = "-" ]; then unset RDSP_X_TAIL_STREAM_PRIO; else export RDSP_X_TAIL_STREAM_PRIO=$4; fi
RDSP_FIR_VARIANT=$2 RDSP_PRIO=$3 python bench.py --config $1 --steps 40 --warmup 10 --no-cpu-baseline --no-host-io --no-extra-legs --no-iso > gpurun_out/pr.json 2>gpurun_out/pr.err || tail -3 gpurun_out/pr.err
python - "$1" "$2" "$3" "$4" <<PY
import json,sys
d=json.loads(open("gpurun_out/pr.json").read().strip().splitlines()[-1])
print(sys.argv[1], "fir_variant", sys.argv[2], "wave prio(front FIR, tail)", sys.argv[3], "tail stream prio", sys.argv[4], "ms/step %.3f steady %.3f"%(d["ms_per_step"], d["ms_per_step_steady"]), {k:round(v,3) for k,v in d["kernels_ms_per_step"].items()})
PY
done; done; done
