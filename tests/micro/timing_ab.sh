#!/bin/bash
# round 4: cost of the per-kernel timing inside the timed region, two ways -- marker events recorded around the
# launches (ab/rec.so: the product) and the launches' own timestamps taken into the events by hipExtLaunchKernel
# (ab/ext.so: a build with rdsp_launch_*_ev, not kept) -- each with and without --no-kernel-timing.  Result:
# 1.149-1.152 / 1.146-1.149 ms timed, 1.133-1.141 un-timed.  Needs the two libraries under ab/.
for rep in 1 2 3; do for lib in ab/rec.so ab/ext.so; do for fl in "" "--no-kernel-timing"; do
python bench.py --lib $PWD/$lib --config K3 --steps 100 --warmup 10 --no-cpu-baseline --no-host-io --no-extra-legs --no-iso $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib [$fl] ms/step %.4f steady %s kernels %s'%(d['ms_per_step'], d['ms_per_step_steady'], d['kernels_ms_per_step']))"
done; done; done
