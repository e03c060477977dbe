# 25 000 K3 steps with clock, package power and temperatures sampled once a second (is the step time steady?)
( while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (junction|memory)" | tr '\n' ' '; echo; sleep 1; done ) > gpurun_out/long_samples.txt &
SP=$!
python bench.py --config K3 --steps 25000 --warmup 20 --no-cpu-baseline --no-host-io --no-iso --no-kernel-timing > gpurun_out/long_k3.json 2> gpurun_out/long_k3.err
kill $SP; wait $SP 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/long_k3.json').read().strip().splitlines()[-1]); print('K3 25000 steps: ms/step %.4f'%d['ms_per_step'])"
grep -o "(\([0-9]*\)Mhz)\|Power (W): [0-9.]*\|(C): [0-9.]*" gpurun_out/long_samples.txt | paste - - - - 2>/dev/null | awk 'NR%3==1' | head -14
