"""A/B of builds (development tool): python tools/ab_bench.py libA.so libB.so[:VAR=value,...] ...
(a suffix sets environment variables for that leg, e.g. rsoccer_amd/librsx_hip.so:RSX_EPL_ZIGZAG=0; LARGE=1 in the
environment restricts the list to the large-batch cases)
Per build: VSS-v0 at 4096 / 65536 / 1 M / 4 M envs (per-step launches and one launch), the four SSL
tasks at 2048 envs and the raw 11v11 step at 1024 — each build in its own process."""
import os
import subprocess
import sys

CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
def leg(sim, n, warm):
    out = []
    for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
        fn(warm); torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / n * 1e6)
    return out
for name, kind, ft, nb, ny, task, B, n, warm in (("vss", 0, 0, 3, 3, 1, 4096, 4000, 2000), ("vss", 0, 0, 3, 3, 1, 65536, 300, 300),
        ("vss", 0, 0, 3, 3, 1, 1 << 20, 100, 100), ("vss", 0, 0, 3, 3, 1, 1 << 22, 60, 60),
        ("sd", 1, 2, 1, 6, 2, 2048, 2000, 300), ("sd", 1, 2, 1, 6, 2, 262144, 100, 100), ("sd", 1, 2, 1, 6, 2, 1 << 20, 60, 60), ("drib", 1, 2, 1, 4, 3, 1 << 20, 60, 60), ("scrim", 1, 1, 11, 11, 6, 131072, 60, 40), ("scrim", 1, 1, 11, 11, 6, 262144, 40, 40), ("scrimC", 1, 1, 11, 11, 7, 262144, 40, 40), ("cont", 1, 2, 1, 1, 4, 1 << 20, 60, 60), ("pass", 1, 2, 2, 0, 5, 1 << 20, 60, 60), ("drib", 1, 2, 1, 4, 3, 2048, 2000, 300),
        ("cont", 1, 2, 1, 1, 4, 2048, 2000, 300), ("pass", 1, 2, 2, 0, 5, 2048, 2000, 300),
        ("scrim", 1, 1, 11, 11, 6, 1024, 2000, 300), ("scrimC", 1, 1, 11, 11, 7, 1024, 2000, 300), ("scrim", 1, 1, 11, 11, 6, 65536, 100, 50),
        ("scrimC", 1, 1, 11, 11, 7, 65536, 100, 50), ("vss5", 0, 1, 5, 5, 1, 4096, 2000, 300)):
    if os.environ.get("LARGE") and B < 100000: continue
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
    a, b = leg(sim, n, warm)
    print(f"{name:5s} {B:8d} step {a:8.2f} us  one-launch {b:8.2f} us/step", flush=True)
    sim.close()
if os.environ.get("LARGE"): sys.exit(0)
sim = L.Sim(1, 1, 11, 11, 25, 1024)
sim.step_dev(s); torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(2000): sim.step_dev(s)
torch.cuda.synchronize(); print(f"11v11     1024 raw  {(time.perf_counter() - t) / 2000 * 1e6:8.2f} us", flush=True)
'''
for arg in sys.argv[1:]:
    lib, _, sets = arg.partition(":")
    extra = dict(kv.split("=", 1) for kv in sets.split(",") if kv)
    print("==", arg, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LIB=os.path.abspath(lib), **extra))
