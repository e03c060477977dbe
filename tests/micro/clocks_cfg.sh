#!/bin/bash
# GPU clock and package power while a bench configuration runs (rocm-smi sampled in the background):
#   bash tests/micro/clocks_cfg.sh K2 K3 K4 F1 [K3np = K3 un-pipelined]
for K in "$@"; do
  F=""; C=$K; if [ "$K" = "K3np" ]; then C=K3; F="--no-pipeline"; fi
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clk_samples_$K.txt &
  SP=$!
  python bench.py --config $C $F --steps 4000 --warmup 20 --no-cpu-baseline --no-host-io --no-iso --no-kernel-timing > gpurun_out/clk_bench_$K.json 2> gpurun_out/clk_bench_$K.err
  kill $SP; wait $SP 2>/dev/null
  python - "$K" <<'PY'
import json, re, sys
K = sys.argv[1]
d = json.loads(open(f"gpurun_out/clk_bench_{K}.json").read().strip().splitlines()[-1])
s = open(f"gpurun_out/clk_samples_{K}.txt").read()
pairs = [(int(a), float(b)) for a, b in re.findall(r"sclk clock level: \S+ \((\d+)Mhz\).*?Power \(W\): ([0-9.]+)", s)]
busy = [(c, p) for c, p in pairs if p > 500]
if busy:
    print(K, "ms/step %.3f" % d["ms_per_step"], "under load: %d samples, clock %.0f MHz (min %d max %d), package %.0f W (max %.0f)" % (
        len(busy), sum(c for c, _ in busy) / len(busy), min(c for c, _ in busy), max(c for c, _ in busy),
        sum(p for _, p in busy) / len(busy), max(p for _, p in busy)))
else:
    print(K, "ms/step %.3f" % d["ms_per_step"], "no sample under load", pairs[:5])
PY
done
