"""Counts the memory copies that fall between the first and the last raw-step kernel of a rocprofv3 run
(memory-copy trace + kernel trace CSVs): the copies issued by step() itself.
usage: python tools/memcopy_window.py <memory_copy_trace.csv> <dir with kernel_trace.csv>"""
import csv, glob, os, sys
mc, d = sys.argv[1], sys.argv[2]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
steps = [r for f in kt for r in csv.DictReader(open(f)) if "sim_step_kernel" in r.get("Kernel_Name", "")]
if not steps:
    print("no sim_step_kernel launches found"); sys.exit(0)
t0 = min(int(r["Start_Timestamp"]) for r in steps); t1 = max(int(r["End_Timestamp"]) for r in steps)
rows = list(csv.DictReader(open(mc)))
inside = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= t1]
kinds = {}
for r in inside:
    kinds[r.get("Direction", "?")] = kinds.get(r.get("Direction", "?"), 0) + 1
print(f"raw-step kernel launches: {len(steps)}; memory copies between the first and the last of them: {len(inside)} {kinds}")
