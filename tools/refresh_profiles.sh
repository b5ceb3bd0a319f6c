#!/bin/bash
# Runs on the MI355X box (via gpurun): regenerates every artefact kept under profiles/.
# usage: tools/refresh_profiles.sh <tag>     (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -1 $OUT/bench.json | cut -c1-400
# kernel trace + stats of the bench command (step mode only, no CPU leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --no-cpu-baseline --no-rollout > $OUT/stats.log 2>&1
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats_pooled.csv
# the same trace grouped by (kernel, grid size): the 4096-env launches of the headline kernel get their own row
python tools/kernel_stats_by_grid.py $OUT/stats $OUT/bench_kernel_stats_by_grid.csv && cp $OUT/bench_kernel_stats_by_grid.csv $OUT/bench_kernel_stats.csv
head -3 $OUT/bench_kernel_stats.csv
# HBM traffic counters, one pass each, no tracing
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-rollout --no-extra > $OUT/pmc_$c.log 2>&1   # (--no-extra: counter collection does not survive the graph-capture legs)
done
python tools/pmc_summary.py $OUT > $OUT/hbm_traffic.json
cat $OUT/hbm_traffic.json | head -12
# SQ counters of one 200-step rollout launch
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/sq -- python tools/prof_target.py 4096 rollout 200 > $OUT/sq.log 2>&1
python tools/rocpd_summary.py $(find $OUT/sq -name "*.db" | head -1) > $OUT/sq_counters_rollout200.txt 2>&1
grep "task_step_kernel<0, 8, 1, 6, 3>" $OUT/sq_counters_rollout200.txt | head -12
# counter traffic of every large-batch leg (bench.py reads profiles/rNN_leg_traffic.json)
tools/prof_leg_traffic.sh $OUT > $OUT/leg_traffic.log 2>&1
python tools/bench_configs.py $OUT/configs_and_sweep.md > $OUT/configs.log 2>&1
cat $OUT/configs_and_sweep.md
# the driver's own flags
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err
# one-lane-per-env kernel at 1 M envs: SQ + HBM counters
tools/prof_epl.sh $OUT/epl > $OUT/epl.log 2>&1
cat $OUT/epl/*.txt > $OUT/epl_counters_1M.txt
# the four registered SSL tasks at 1 M envs (one-lane-per-env kernels): kernel stats, SQ + HBM counters of the 1v6 one
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ssl_stats -- python -c "
import sys; sys.path.insert(0, '.')
import torch
from rsoccer_amd import _lib as L
for task, nb, ny in ((2, 1, 6), (3, 1, 4), (4, 1, 1), (5, 2, 0)):
    s = L.Sim(1, 2, nb, ny, 25, 1 << 20); s.task_attach(task, 0, 0, 0); s.task_reset(); s.task_step_n(60); torch.cuda.synchronize(); s.close()
" > $OUT/ssl_stats.log 2>&1
find $OUT/ssl_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/ssl_tasks_1M_kernel_stats.csv
head -6 $OUT/ssl_tasks_1M_kernel_stats.csv
tools/prof_kernel.sh $OUT/sd_epl 1048576 2 "ssl_epl_kernel" > $OUT/sd_epl.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $OUT/sd_epl/$c -- python tools/prof_target.py 1048576 step 12 2 200 > $OUT/sd_epl/$c.log 2>&1
  python tools/rocpd_summary.py $(find $OUT/sd_epl/$c -name "*.db" | head -1) 2>&1 | grep -E "ssl_epl_kernel" | grep -v "^rsx" > $OUT/sd_epl/$c.txt
done
cat $OUT/sd_epl/*.txt > $OUT/static_defenders_epl_counters_1M.txt
cut -c1-130 $OUT/static_defenders_epl_counters_1M.txt
# SSL 11v11 scrimmage: 32 lanes per env against four lanes per env over the batch sizes; SQ counters of the four-lane kernel at 262 144 envs
python tools/quick_scrim.py 2>&1 | grep -v amdgpu > $OUT/scrimmage_layouts.txt
cat $OUT/scrimmage_layouts.txt
tools/prof_kernel.sh $OUT/quad 262144 6 "ssl_quad_kernel" > $OUT/quad.log 2>&1
cat $OUT/quad/*.txt > $OUT/scrimmage_quad_counters_262144.txt
cut -c1-130 $OUT/scrimmage_quad_counters_262144.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
# batched hooks: no host <-> device copy inside step()
rocprofv3 --memory-copy-trace --kernel-trace --stats --output-format csv -d $OUT/hooks -- python tools/hooks_nocopy.py > $OUT/hooks.log 2>&1
find $OUT/hooks -name "*memory_copy_stats.csv" -o -name "*memory_copy_trace.csv" | head -3
( echo "# rocprofv3 --memory-copy-trace -- python tools/hooks_nocopy.py (4096 envs, 300 steps of a hook-written task)"; grep -E "STEPS" $OUT/hooks.log;
  for f in $(find $OUT/hooks -name "*memory_copy_trace.csv" | head -1); do echo "memory copies in the whole run: $(($(wc -l < $f) - 1))"; python tools/memcopy_window.py $f $OUT/hooks; done ) > $OUT/hooks_memcopy.txt 2>&1
cat $OUT/hooks_memcopy.txt
# in-kernel timeline of the headline kernel
tools/build_timing.sh > /dev/null 2>&1 && RSX_LIB=tools/_dev/librsx_hip_timing.so python tools/exp_timeline2.py > $OUT/timeline.txt 2>&1
head -40 $OUT/timeline.txt
for t in 2 3 5; do echo "== task $t, 1 048 576 envs, one lane per env"; RSX_LIB=tools/_dev/librsx_hip_timing.so B=1048576 TASK=$t python tools/exp_timeline_epl.py 2>&1 | grep -v amdgpu; done > $OUT/timeline_ssl_epl_1M.txt
cat $OUT/timeline_ssl_epl_1M.txt
