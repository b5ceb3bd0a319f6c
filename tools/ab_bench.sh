#!/bin/bash
# A/B of several builds of librsx_hip on the same box, interleaved (development tool).
# usage: tools/ab_bench.sh rounds lib1.so lib2.so ...
R=$1; shift
for i in $(seq $R); do
  for lib in "$@"; do
    echo "== $lib"
    RSX_LIB=$lib python tools/quick_bench.py ${QB_ARGS:-4096} 2>&1 | grep "B="
  done
done
