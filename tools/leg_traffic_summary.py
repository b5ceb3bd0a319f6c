"""Summarises the per-leg FETCH_SIZE / WRITE_SIZE passes of tools/prof_leg_traffic.sh: per leg the median counter value of
the per-step kernel's dispatches, in KB, scaled by the factors tools/prof_calibration.sh measured on launches of the same kernels with a known byte count
(profiles/rNN_counter_calibration.json; MI355X_MICROARCH.md, HBM section, asks for exactly that calibration)."""
import csv, glob, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
root = sys.argv[1]
from calib_factors import calibrated_factors

FF, WF, FSRC = calibrated_factors()
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) -- python tools/leg_target.py <leg>; "
                 "per-step launches after a warm-up launch; MI355X",
       "correction": f"KB counters; bytes_per_launch = {FF:g} * FETCH_SIZE + {WF:g} * WRITE_SIZE, factors measured on launches of these kernels with a known byte count: {FSRC}",
       "factors": {"FETCH_SIZE": FF, "WRITE_SIZE": WF, "source": FSRC},
       "legs": {}}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    leg = os.path.basename(d).replace("_", ":")
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals, names = {}, {}
        for f in glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") != c:
                    continue
                k = r.get("Kernel_Name", "")
                if any(s in k for s in ("rollout", "fold_metrics", "fillBuffer", "copyBuffer", "reset_dev")):
                    continue
                vals.setdefault(k, []).append(float(r["Counter_Value"]))
        if not vals:
            continue
        # the per-step kernel: the name with the most dispatches (reset and the warm-up launch are one dispatch each)
        k = max(vals, key=lambda n: len(vals[n]))
        rec["kernel"] = k.split("(")[0]
        rec[c + "_KB"] = statistics.median(vals[k])
        rec["dispatches"] = len(vals[k])
    if "FETCH_SIZE_KB" in rec and "WRITE_SIZE_KB" in rec:
        rec["bytes_per_launch"] = int((FF * rec["FETCH_SIZE_KB"] + WF * rec["WRITE_SIZE_KB"]) * 1024)
    out["legs"][leg] = rec
print(json.dumps(out, indent=1))
