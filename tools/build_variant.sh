#!/bin/bash
# Development build of librsx_hip.so with extra compiler flags -> tools/_dev/librsx_<name>.so
# (same per-unit flags as __graft_entry__.build()):  tools/build_variant.sh <name> [-DFOO=1 ...]
# Use with RSX_LIB=tools/_dev/librsx_<name>.so (tools/ab_configs.sh, bench.py, tests).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_dev
C="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=12 -Iinclude -Irsoccer_amd/csrc $* -c"
hipcc $C -mllvm -amdgpu-sched-strategy=max-ilp -o /tmp/_rsx_${name}_api.o rsoccer_amd/csrc/rsx_api.hip &
hipcc $C -fno-slp-vectorize -o /tmp/_rsx_${name}_epl.o rsoccer_amd/csrc/rsx_epl.hip &
hipcc $C -fno-slp-vectorize -o /tmp/_rsx_${name}_big.o rsoccer_amd/csrc/rsx_big.hip &
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o tools/_dev/librsx_${name}.so /tmp/_rsx_${name}_api.o /tmp/_rsx_${name}_epl.o /tmp/_rsx_${name}_big.o
echo built tools/_dev/librsx_${name}.so
