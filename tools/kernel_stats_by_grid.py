"""Per-kernel timing statistics of a `rocprofv3 --kernel-trace --output-format csv` run, grouped by (kernel, grid size): launches of
one kernel symbol at different batch sizes (bench.py steps VSS-v0 at 4096 AND 65 536 envs with the same task_step_kernel<0,8,1,6,0>)
get their own rows — rocprofv3's own --stats table pools them.
    python tools/kernel_stats_by_grid.py <dir-or-kernel_trace.csv> [out.csv]
Columns follow rocprofv3's kernel_stats.csv plus Grid_Size (work-items; workgroups = Grid_Size / 64 for every kernel of this library)."""
import csv
import glob
import os
import sys


def find_trace(path):
    if os.path.isfile(path):
        return path
    hits = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
    if not hits:
        raise SystemExit(f"no *kernel_trace.csv under {path}")
    return hits[0]


def main():
    src = find_trace(sys.argv[1])
    groups = {}
    with open(src, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            name = r.get("Kernel_Name") or r.get("Name")
            gx = int(r.get("Grid_Size_X") or r.get("Grid_Size") or 0)
            gy, gz = int(r.get("Grid_Size_Y") or 1), int(r.get("Grid_Size_Z") or 1)
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            groups.setdefault((name, gx * max(gy, 1) * max(gz, 1)), []).append(d)
    total = sum(sum(v) for v in groups.values()) or 1
    rows = []
    for (name, grid), v in groups.items():
        n, s = len(v), sum(v)
        mean = s / n
        var = sum((x - mean) ** 2 for x in v) / n
        rows.append((name, grid, n, s, mean, 100.0 * s / total, min(v), max(v), var ** 0.5))
    rows.sort(key=lambda r: -r[3])
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Grid_Size", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3], round(r[4], 3), round(r[5], 4), r[6], r[7], round(r[8], 3)])


if __name__ == "__main__":
    main()
