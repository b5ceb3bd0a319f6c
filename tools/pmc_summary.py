"""Summarises the FETCH_SIZE / WRITE_SIZE passes of tools/refresh_profiles.sh into the JSON that
bench.py reads for roofline.traffic (factors: tools/calib_factors.py)."""
import csv, glob, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from calib_factors import calibrated_factors
FF, WF, FSRC = calibrated_factors()
out = sys.argv[1]
KERNEL = "task_step_kernel<0, 8, 1, 6, 0>"
GRID = 512 * 64   # work-items of a 4096-env launch (8 envs per 64-thread workgroup)
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = []
    for f in glob.glob(os.path.join(out, f"pmc_{c}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            # only the 4096-env launches of the symbol (512 workgroups of 64): the same kernel also steps the 65 536-env sweep leg
            if KERNEL in r.get("Kernel_Name", "") and r.get("Counter_Name") == c and int(r.get("Grid_Size", GRID) or GRID) == GRID:
                rows.append(float(r["Counter_Value"]))
    vals[c] = statistics.median(rows) if rows else None
B = 4096
res = {
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, no tracing) -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-rollout, MI355X",
    "kernel": f"rsx::{KERNEL} (one launch = {B} envs x 1 fused VSS-v0 step)",
    "FETCH_SIZE_KB_median_per_launch": vals["FETCH_SIZE"],
    "WRITE_SIZE_KB_median_per_launch": vals["WRITE_SIZE"],
    "correction": f"counters are in KB; bytes_per_launch = {FF:g} * FETCH_SIZE + {WF:g} * WRITE_SIZE, factors measured on launches of this kernel family with a known byte count: {FSRC}",
    "factors": {"FETCH_SIZE": FF, "WRITE_SIZE": WF, "source": FSRC},
    "bytes_per_launch": None if None in vals.values() else int(FF * vals["FETCH_SIZE"] * 1024 + WF * vals["WRITE_SIZE"] * 1024),
    "expected_from_layout": {"read_B_per_env": 252, "write_B_per_env": 418,
                             "note": "state 44 f32 (36 robot + 5 ball + height, vz, spin) + steps/episode + OU 10 + info 6 + prev_pot read; the same + obs 40 f32 + reward + 2 flag bytes written"},
    "algorithmic_bytes_per_launch": 541 * B,
}
# the rocprofv3 --kernel-trace average of the SAME launches (tools/kernel_stats_by_grid.py), so that roofline.frac can be recomputed from one row
by_grid = os.path.join(out, "bench_kernel_stats_by_grid.csv")
if os.path.exists(by_grid):
    for r in csv.DictReader(open(by_grid)):
        if KERNEL in r["Name"] and int(float(r["Grid_Size"])) == GRID:
            avg_us = float(r["AverageNs"]) / 1000.0
            res["rocprof_kernel_trace_4096_envs"] = {"calls": int(float(r["Calls"])), "avg_launch_us": round(avg_us, 4), "min_us": float(r["MinNs"]) / 1000.0,
                                                   "max_us": float(r["MaxNs"]) / 1000.0,
                                                   "algorithmic_GBps": round(541 * B / avg_us / 1e3, 2), "frac_of_8TBps": round(541 * B / avg_us / 1e3 / 8000.0, 5),
                                                   "source": "bench_kernel_stats_by_grid.csv (rocprofv3 --kernel-trace of `python bench.py --no-cpu-baseline --no-rollout`, grouped by kernel and grid size)"}
            break
print(json.dumps(res, indent=1))
