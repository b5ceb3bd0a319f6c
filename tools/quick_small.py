"""Development: per-step launch time of the latency-bound SSL configurations for a list of library builds (each in its own process,
two rounds): python tools/quick_small.py libA.so libB.so ..."""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
out = []
for name, kind, ft, nb, ny, task, B in (("sd", 1, 2, 1, 6, 2, 2048), ("cont", 1, 2, 1, 1, 4, 2048), ("scrim", 1, 1, 11, 11, 6, 1024), ("scrimC", 1, 1, 11, 11, 7, 1024), ("vss", 0, 0, 3, 3, 1, 4096)):
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
    sim.task_step_n(500, s); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); sim.task_step_n(2000, s); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 2000 * 1e6)
    out.append(f"{name} {best:6.2f}")
    sim.close()
sim = L.Sim(1, 1, 11, 11, 25, 1024)
sim.step_dev(s); torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    t = time.perf_counter()
    for _ in range(2000): sim.step_dev(s)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 2000 * 1e6)
out.append(f"raw11 {best:6.2f}")
print("  ".join(out), flush=True)
'''
for rnd in range(2):
    for lib in sys.argv[1:]:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LIB=lib), capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
