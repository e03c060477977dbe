#!/bin/bash
# s_memtime phases of the tail kernel (cycles per 128-step block: top | NLMS block() | ring rotation | AGC, pack):
# builds a copy of the library with -DRDSP_TAIL_PROFILE into variants/prof and runs tests/micro/tail_bench on it.
# usage: bash tests/micro/tail_phases.sh   (build here, run the printed command on the GPU box)
set -e
cd "$(dirname "$0")/../../radiodsp_sdr_rx_amd/csrc"
make > /dev/null
mkdir -p ../../variants/prof
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-slp-vectorize -DRDSP_TAIL_PROFILE -c rdsp_tail.hip -o /tmp/rdsp_tail_prof.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../variants/prof/librdsp_hip.so rdsp_kernels.o /tmp/rdsp_tail_prof.o rdsp_sam.o rdsp_chain.o rdsp_design.o rdsp_graph.o rdsp_graph_sdr.o rdsp_spectrum.o rdsp_biquad.o rdsp_fft1024.o rdsp_io.o rdsp_stream.o -fopenmp -lm
echo 'gpurun -- "LD_LIBRARY_PATH=\$PWD/variants/prof tests/micro/tail_bench 4096 100"'
