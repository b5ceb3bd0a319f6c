#!/bin/bash
# On the GPU box: HBM traffic counters of the one-lane-per-env kernel at 1 M envs (per-step launches) for a given library
OUT=${1:-gpurun_out/epl_traffic}; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$c
  rocprofv3 --pmc $c -d $OUT/$c -- python tools/prof_target.py 1048576 step 12 > $OUT/$c.log 2>&1
  python tools/rocpd_summary.py $(find $OUT/$c -name "*.db" | head -1) 2>&1 | grep -E "vss_epl_kernel" | grep "$c" | awk '{print $(NF-2), $(NF)}'
done
