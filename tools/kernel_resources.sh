#!/bin/bash
# Prints VGPR / SGPR / scratch / LDS / occupancy per kernel of librsx_hip (hipcc remarks), with the
# per-unit flags of __graft_entry__.HIP_UNITS.
cd "$(dirname "$0")/.."
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -mllvm -amdgpu-kernarg-preload-count=12 -Iinclude -Irsoccer_amd/csrc -Rpass-analysis=kernel-resource-usage -c"
( hipcc $COMMON -mllvm -amdgpu-sched-strategy=max-ilp -o /tmp/_rsx_probe_api.o rsoccer_amd/csrc/rsx_api.hip 2>&1
  hipcc $COMMON -fno-slp-vectorize -o /tmp/_rsx_probe_epl.o rsoccer_amd/csrc/rsx_epl.hip 2>&1
  hipcc $COMMON -fno-slp-vectorize -o /tmp/_rsx_probe_big.o rsoccer_amd/csrc/rsx_big.hip 2>&1 ) | python3 -c '
import sys,re
cur=None;rows={}
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r"remark:\s+([A-Za-z ]+?)(?: \[.*?\])?: (\d+)",l)
    if m and cur: rows[cur][m.group(1).strip()]=m.group(2)
print("%-70s %5s %5s %7s %6s %4s"%("kernel","VGPR","SGPR","scratch","LDS","occ"))
for k,v in rows.items():
    print("%-70s %5s %5s %7s %6s %4s"%(k[:70],v.get("VGPRs"),v.get("TotalSGPRs"),v.get("ScratchSize"),v.get("LDS Size"),v.get("Occupancy")))
'
