#!/bin/bash
# Runs on the MI355X box (via gpurun): regenerates every artefact kept under profiles/.
# usage: tools/refresh_profiles.sh <tag>     (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
# kernel trace + stats of the bench command (step mode only, no CPU leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --no-cpu-baseline --no-rollout > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
head -3 $OUT/bench_kernel_stats.csv
# HBM traffic counters, one pass each, no tracing
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-rollout > $OUT/pmc_$c.log 2>&1
done
python tools/pmc_summary.py $OUT > $OUT/hbm_traffic.json
cat $OUT/hbm_traffic.json | head -12
# SQ counters of one 200-step rollout launch
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -- python tools/prof_target.py 4096 rollout 200 > $OUT/sq.log 2>&1
python tools/rocpd_summary.py $(find $OUT/sq -name "*.db" | head -1) > $OUT/sq_counters_rollout200.txt 2>&1
grep "task_step_kernel<0, 8, 1, 6, 3>" $OUT/sq_counters_rollout200.txt | head -12
python tools/bench_configs.py $OUT/configs_and_sweep.md > $OUT/configs.log 2>&1
cat $OUT/configs_and_sweep.md
