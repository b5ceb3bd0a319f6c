#!/bin/bash
# On the GPU box: SQ counters of the step kernel of one task at one batch size (per-step launches).
# usage: tools/prof_kernel.sh <outdir> <envs> <task id> [kernel name pattern]
OUT=${1:-gpurun_out/kprof}; B=${2:-65536}; TASK=${3:-6}; PAT=${4:-epl_kernel}; mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $OUT/$tag -- python tools/prof_target.py $B step 12 $TASK 200 > $OUT/$tag.log 2>&1
  python tools/rocpd_summary.py $(find $OUT/$tag -name "*.db" | head -1) 2>&1 | grep -E "$PAT" | grep -v "^rsx" > $OUT/$tag.txt
  cat $OUT/$tag.txt | cut -c1-130
done
