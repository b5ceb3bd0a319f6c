#!/bin/bash
# Development build with in-kernel time stamps (-DRSX_TIMING) -> tools/_dev/librsx_hip_timing.so,
# read by tools/exp_timeline2.py (RSX_LIB=tools/_dev/librsx_hip_timing.so).  Same flags per
# translation unit as __graft_entry__.build().
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_dev
C="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=12 -Iinclude -Irsoccer_amd/csrc -DRSX_TIMING -c"
hipcc $C -mllvm -amdgpu-sched-strategy=max-ilp -o /tmp/_rsx_t_api.o rsoccer_amd/csrc/rsx_api.hip
hipcc $C -fno-slp-vectorize -o /tmp/_rsx_t_epl.o rsoccer_amd/csrc/rsx_epl.hip
hipcc $C -fno-slp-vectorize -o /tmp/_rsx_t_big.o rsoccer_amd/csrc/rsx_big.hip
hipcc --offload-arch=gfx950 -fPIC -shared -o tools/_dev/librsx_hip_timing.so /tmp/_rsx_t_api.o /tmp/_rsx_t_epl.o /tmp/_rsx_t_big.o
echo built tools/_dev/librsx_hip_timing.so
