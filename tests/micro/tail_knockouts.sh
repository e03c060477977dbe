#!/bin/bash
# Knock-out table of the product tail kernel (wrong results, valid timing): variants/ko/lib_<mask>.so are copies
# of the library whose rdsp_tail.hip was compiled with pieces of the two-step block removed (bit mask:
# 1 reduction, 2 broadcast-subtracts, 4 error / step size, 8 five of six updates, 16 five of six dot products,
# 32 pair reads, 64 scalar reads, 128 prefix scans, 256 output stores, 512 AGC).  Built by the recipe in
# DESIGN.md 4.2's history (a patched copy of rdsp_tail.hip, not kept in the tree); run on the GPU box:
#   bash tests/micro/tail_knockouts.sh
for ko in 0 1 2 3 4 8 16 32 64 128 256 512 27 31 1023; do
  mkdir -p /tmp/ko_$ko && cp variants/ko/lib_$ko.so /tmp/ko_$ko/librdsp_hip.so
  a=$(LD_LIBRARY_PATH=/tmp/ko_$ko tests/micro/tail_bench 4096 100 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  b=$(LD_LIBRARY_PATH=/tmp/ko_$ko tests/micro/tail_bench 4096 100 | tail -1 | sed 's/.*: \([0-9.]*\) ms.*/\1/')
  echo "KO $ko: $a $b ms"
done
