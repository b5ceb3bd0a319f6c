"""Summarise a rocprofv3 rocpd sqlite database: per-kernel timing stats and per-kernel PMC sums.
usage: python tools/rocpd_summary.py <results.db> [...]"""
import sqlite3, sys

def cols(c, t):
    return [r[1] for r in c.execute(f"pragma table_info({t})")]

for f in sys.argv[1:]:
    c = sqlite3.connect(f)
    print("==", f)
    kc = cols(c, "kernels")
    name = "name" if "name" in kc else "kernel_name"
    try:
        rows = c.execute(f"select {name}, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by {name} order by 6 desc").fetchall()
        print("%-72s %7s %10s %10s %10s %12s" % ("kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ns"))
        for r in rows:
            print("%-72s %7d %10.0f %10d %10d %12d" % (str(r[0])[:72], r[1], r[2], r[3], r[4], r[5]))
    except Exception as e:
        print("kernels view:", e, kc)
    try:
        pc = cols(c, "counters_collection")
        kn = "kernel_name" if "kernel_name" in pc else "name"
        rows = c.execute(f"select {kn}, counter_name, count(*), avg(value), sum(value) from counters_collection group by {kn}, counter_name order by 1,2").fetchall()
        if rows:
            print("%-50s %-24s %7s %16s" % ("kernel", "counter", "n", "avg_per_dispatch"))
            for r in rows:
                print("%-50s %-24s %7d %16.1f" % (str(r[0])[:50], r[1], r[2], r[3]))
    except Exception as e:
        print("counters view:", e, pc)
