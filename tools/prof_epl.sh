#!/bin/bash
# On the GPU box: hardware counters of the one-lane-per-env kernel at 1 M envs, per-step launches.
# (12 launches after 300 steps of warm-up in one launch: steady-state steps, contacts included)
# usage: tools/prof_epl.sh <outdir>
OUT=${1:-gpurun_out/epl_prof}; mkdir -p $OUT; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set -d $OUT/$tag -- python tools/prof_target.py 1048576 step 12 1 300 > $OUT/$tag.log 2>&1
  python tools/rocpd_summary.py $(find $OUT/$tag -name "*.db" | head -1) 2>&1 | grep -E "vss_epl_kernel" | grep -v "^rsx" > $OUT/$tag.txt
  cat $OUT/$tag.txt | cut -c1-120
done
