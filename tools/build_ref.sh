#!/bin/bash
# Development: build librsx_hip from the sources of a git ref -> tools/_dev/librsx_<name>.so (A/B against the working tree)
#   tools/build_ref.sh <name> <git-ref> [-DFOO ...]
set -e
cd "$(dirname "$0")/.."
name=$1; ref=$2; shift 2
src=tools/_dev/src_$name
rm -rf $src; mkdir -p $src/csrc $src/include
for f in $(git ls-tree --name-only $ref rsoccer_amd/csrc/); do git show $ref:$f > $src/csrc/$(basename $f); done
git show $ref:include/rsx.h > $src/include/rsx.h
C="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=12 -I$src/include -I$src/csrc $* -c"
hipcc $C -mllvm -amdgpu-sched-strategy=max-ilp -o /tmp/_rsx_${name}_api.o $src/csrc/rsx_api.hip &
hipcc $C -fno-slp-vectorize -o /tmp/_rsx_${name}_epl.o $src/csrc/rsx_epl.hip &
hipcc $C -fno-slp-vectorize -o /tmp/_rsx_${name}_big.o $src/csrc/rsx_big.hip &
wait
hipcc --offload-arch=gfx950 -fPIC -shared -o tools/_dev/librsx_${name}.so /tmp/_rsx_${name}_api.o /tmp/_rsx_${name}_epl.o /tmp/_rsx_${name}_big.o
echo built tools/_dev/librsx_${name}.so
