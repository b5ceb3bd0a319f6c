#!/bin/bash
# On the GPU box: only the HBM-traffic counters of the headline kernel (-> <out>/hbm_traffic.json)
OUT=${1:-gpurun_out/r02}; mkdir -p $OUT; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-rollout --no-extra > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/hbm_traffic.json
cat $OUT/hbm_traffic.json
