"""Development tools that produce committed evidence get a functional check of their own (CPU)."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_stats_by_grid_separates_batch_sizes_of_one_symbol(tmp_path):
    """tools/kernel_stats_by_grid.py: launches of ONE kernel symbol at two grid sizes (bench.py steps VSS-v0 at 4096 and at 65 536 envs
    with the same task_step_kernel<0,8,1,6,0>) end up in two rows with their own averages — rocprofv3's --stats table pools them"""
    trace = tmp_path / "x_kernel_trace.csv"
    name = "void rsx::task_step_kernel<0, 8, 1, 6, 0>(float*)"
    with open(trace, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"])
        t = 1000
        for i in range(10):
            w.writerow(["KERNEL_DISPATCH", name, t, t + 9000 + i, 32768, 1, 1]); t += 20000
        for i in range(4):
            w.writerow(["KERNEL_DISPATCH", name, t, t + 25000, 524288, 1, 1]); t += 40000
        w.writerow(["KERNEL_DISPATCH", "other_kernel()", t, t + 500, 64, 1, 1])
    out = tmp_path / "by_grid.csv"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "kernel_stats_by_grid.py"), str(tmp_path), str(out)])
    rows = {(r["Name"], int(float(r["Grid_Size"]))): r for r in csv.DictReader(open(out))}
    assert len(rows) == 3
    small, big = rows[(name, 32768)], rows[(name, 524288)]
    assert int(float(small["Calls"])) == 10 and abs(float(small["AverageNs"]) - 9004.5) < 1e-6
    assert int(float(big["Calls"])) == 4 and float(big["AverageNs"]) == 25000.0
    assert float(small["MinNs"]) == 9000 and float(small["MaxNs"]) == 9009
