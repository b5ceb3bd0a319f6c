#!/bin/bash
# Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on launches with a known byte count in the step kernels' own
# access patterns (tools/calib_target.py) -> <out>/counter_calibration.json (copied to profiles/rNN_counter_calibration.json; the
# factors are what tools/leg_traffic_summary.py and tools/pmc_summary.py apply).
# usage (on the MI355X box): tools/prof_calibration.sh gpurun_out/<tag>
set -u
OUT=$1; shift
mkdir -p $OUT/calib
export TMPDIR=/tmp
CASES=${*:-"vss:4096:lanes vss:65536:lanes vss:1048576 vss:4194304 sd:1048576 sd:4194304"}
for c in $CASES; do
  for k in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/calib/${c//:/_}/$k
    rm -rf $d; mkdir -p $d
    rocprofv3 --pmc $k --output-format csv -d $d -- python tools/calib_target.py $c > $d/run.log 2>&1
  done
done
python tools/calib_summary.py $OUT/calib > $OUT/counter_calibration.json
cat $OUT/counter_calibration.json
