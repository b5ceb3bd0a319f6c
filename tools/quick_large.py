"""Development: per-step launch time of the large-batch SSL legs for a list of library builds, interleaved rounds (the 1 M-env legs
vary by +-3 % from process to process: compare medians over the rounds): python tools/quick_large.py [rounds] libA.so libB.so ..."""
import os, subprocess, sys, statistics
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
out = []
for name, kind, ft, nb, ny, task, B in (("sd256k", 1, 2, 1, 6, 2, 262144), ("sd1M", 1, 2, 1, 6, 2, 1 << 20), ("drib1M", 1, 2, 1, 4, 3, 1 << 20), ("cont1M", 1, 2, 1, 1, 4, 1 << 20),
                                        ("pass1M", 1, 2, 2, 0, 5, 1 << 20), ("scrC64k", 1, 1, 11, 11, 7, 65536), ("scr256k", 1, 1, 11, 11, 6, 262144)):
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
    sim.task_step_n(60, s); torch.cuda.synchronize()
    best = []
    for _ in range(3):
        t = time.perf_counter(); sim.task_step_n(40, s); torch.cuda.synchronize(); best.append((time.perf_counter() - t) / 40 * 1e6)
    out.append(f"{name} {sorted(best)[1]:7.2f}")
    sim.close()
print("  ".join(out), flush=True)
'''
rounds = int(sys.argv[1]); libs = sys.argv[2:]
acc = {l: [] for l in libs}
for rnd in range(rounds):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LIB=lib), capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        acc[lib].append(line.split())
for lib in libs:
    rows = acc[lib]
    names = rows[0][0::2]
    med = [statistics.median(float(r[2 * i + 1]) for r in rows) for i in range(len(names))]
    print(f"{os.path.basename(lib):24s} " + "  ".join(f"{n} {m:7.2f}" for n, m in zip(names, med)), flush=True)
