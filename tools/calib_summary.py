"""Summarises tools/prof_calibration.sh: per case the median FETCH_SIZE / WRITE_SIZE (KB) of the rollout(0) launches against the
bytes those launches are known to read / write, and the resulting factors (true bytes per counted byte)."""
import csv, glob, json, os, re, statistics, sys
root = sys.argv[1]
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing) -- python tools/calib_target.py <case>: "
                 "rsx_task_rollout(0) launches (load every row, store every row, no step) right after a reset; MI355X",
       "cases": {}}
for d in sorted(glob.glob(os.path.join(root, "*"))):
    case = os.path.basename(d)
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        log = open(os.path.join(d, c, "run.log")).read()
        m = re.search(r"layout (\S+) launches (\d+) read_bytes (\d+) write_bytes (\d+)", log)
        if not m:
            rec["error"] = log[-300:]
            continue
        rec["layout"], rec["read_bytes"], rec["write_bytes"] = m.group(1), int(m.group(3)), int(m.group(4))
        vals = {}
        for f in glob.glob(os.path.join(d, c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "")
                if r.get("Counter_Name") == c and ("rollout" in k or re.search(r", 3(, false)?>", k)):
                    vals.setdefault(k, []).append(float(r["Counter_Value"]))
        if not vals:
            continue
        k = max(vals, key=lambda n: len(vals[n]))
        rec["kernel"] = k.split("(")[0]
        v = vals[k][1:] if len(vals[k]) > 2 else vals[k]   # the first launch follows the reset kernel: cold
        rec[c + "_KB_median"] = statistics.median(v)
        rec[c + "_KB_min_max"] = [min(v), max(v)]
    if "FETCH_SIZE_KB_median" in rec and "WRITE_SIZE_KB_median" in rec:
        rec["fetch_factor"] = rec["read_bytes"] / (rec["FETCH_SIZE_KB_median"] * 1024)
        rec["write_factor"] = rec["write_bytes"] / (rec["WRITE_SIZE_KB_median"] * 1024)
    out["cases"][case] = rec
ff = [r["fetch_factor"] for r in out["cases"].values() if "fetch_factor" in r]
wf = [r["write_factor"] for r in out["cases"].values() if "write_factor" in r]
if ff:
    out["fetch_factor_range"] = [min(ff), max(ff)]
    out["write_factor_range"] = [min(wf), max(wf)]
print(json.dumps(out, indent=1))
