#!/bin/bash
# HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, no tracing) of the per-step launch of every
# large-batch leg bench.py reports -> <out>/leg_traffic.json (copied to profiles/rNN_leg_traffic.json, read by bench.py).
# usage (on the MI355X box): tools/prof_leg_traffic.sh gpurun_out/<tag> [legs...]
set -u
OUT=$1; shift
mkdir -p $OUT/legs
export TMPDIR=/tmp
LEGS=${*:-"vss:65536 vss:1048576 vss:4194304 sd:262144 sd:1048576 sd:4194304 drib:1048576 cont:1048576 pass:1048576 scrim:65536 scrimC:65536 scrim:262144 scrimC:262144"}
for leg in $LEGS; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/legs/${leg/:/_}/$c
    rm -rf $d; mkdir -p $d
    rocprofv3 --pmc $c --output-format csv -d $d -- python tools/leg_target.py $leg > $d/run.log 2>&1
  done
done
python tools/leg_traffic_summary.py $OUT/legs > $OUT/leg_traffic.json
cat $OUT/leg_traffic.json
