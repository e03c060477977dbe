#!/bin/bash
# GPU clock and power while the K3 bench runs (rocm-smi sampled in the background)
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.3; done ) > gpurun_out/clk_samples.txt &
SP=$!
python bench.py --config K3 --steps 6000 --warmup 20 --no-cpu-baseline --no-host-io --no-iso > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err
kill $SP
python -c "import json; d=json.loads(open('gpurun_out/clk_bench.json').read().strip().splitlines()[-1]); print('ms/step %.3f'%d['ms_per_step'])"
grep -o "sclk clock level: [^ ]* ([0-9]*Mhz).*Power (W): [0-9.]*" gpurun_out/clk_samples.txt | sed 's/sclk clock level: //; s/GPU\[0\].*Package //' | sort | uniq -c | sort -rn | head -12
