#!/bin/bash
# clock and package power under the tail kernel alone (tests/micro/tail_bench looped) -- beside clocks_cfg.sh
hipcc -O3 --offload-arch=gfx950 -std=c++17 -I radiodsp_sdr_rx_amd/csrc tests/micro/tail_bench.hip -L radiodsp_sdr_rx_amd -lrdsp_hip -Wl,-rpath,$PWD/radiodsp_sdr_rx_amd -o /tmp/tail_bench || exit 1
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clk_samples_tail.txt &
SP=$!
/tmp/tail_bench 4096 100 12000
kill $SP; wait $SP 2>/dev/null
python - <<'PY'
import re
s = open("gpurun_out/clk_samples_tail.txt").read()
pairs = [(int(a), float(b)) for a, b in re.findall(r"sclk clock level: \S+ \((\d+)Mhz\).*?Power \(W\): ([0-9.]+)", s)]
busy = [(c, p) for c, p in pairs if p > 400]
print("tail alone: %d samples under load, clock %.0f MHz, package %.0f W (max %.0f)" % (len(busy), sum(c for c, _ in busy) / max(1, len(busy)), sum(p for _, p in busy) / max(1, len(busy)), max([p for _, p in busy] or [0])))
PY
