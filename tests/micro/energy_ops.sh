#!/bin/bash
# Energy per lane-operation by instruction kind: tests/micro/energy_ops.hip looped under rocm-smi sampling.
hipcc -O3 --offload-arch=gfx950 -std=c++17 tests/micro/energy_ops.hip -o /tmp/energy_ops 2>/dev/null || { echo build failed; exit 1; }
for M in idle pkfma pkadd fma add dpp cvt ldsr ldsw; do
  ( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/energy_$M.txt &
  SP=$!
  timeout -k 5 60 /tmp/energy_ops $M 4 > gpurun_out/energy_$M.log 2>&1
  kill $SP; wait $SP 2>/dev/null
  python - "$M" <<'PY'
import re, sys
M = sys.argv[1]
s = open(f"gpurun_out/energy_{M}.txt").read()
log = open(f"gpurun_out/energy_{M}.log").read().strip()
pairs = [(int(a), float(b)) for a, b in re.findall(r"sclk clock level: \S+ \((\d+)Mhz\).*?Power \(W\): ([0-9.]+)", s)]
pairs = pairs[len(pairs) // 3:]   # the last two thirds of the run: under load, settled
m = re.search(r"([0-9.e+]+) lane-ops/s", log)
rate = float(m.group(1)) if m else 0.0
clk = sum(c for c, _ in pairs) / max(1, len(pairs)); pw = sum(p for _, p in pairs) / max(1, len(pairs))
print(f"{M:6s} clock {clk:5.0f} MHz  package {pw:6.0f} W  lane-ops/s {rate:.3e}  | {log}")
PY
done
