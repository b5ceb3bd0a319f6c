"""Development: the 11v11 scrimmage task, single-step launches of the four-lanes-per-env kernel (RSX_LAYOUT=quad), three
successive windows of n steps after a reset (the crowded line-up is densest in the first ~80 steps).  RSX_LIB picks the build."""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("RSX_LAYOUT", "quad")
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
NB = 2 * 4 * (5 + 11 * 22) + 4 * 8 * 22 + 4 * 46 + 5
sizes = [int(a) for a in sys.argv[1:]] or [65536, 262144]
for task, name in ((6, "spread"), (7, "crowded")):
    for B in sizes:
        n = 100 if B <= 65536 else 40
        sim = L.Sim(1, 1, 11, 11, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
        out = []
        for rep in range(3):
            torch.cuda.synchronize(); t = time.perf_counter(); sim.task_step_n(n, s); torch.cuda.synchronize()
            out.append((time.perf_counter() - t) / n * 1e6)
        print(f"11v11 {name:8s} {B:8d} {sim.task_layout():20s} windows of {n}: " + "  ".join(f"{o:8.2f} us ({NB*B/o/8e4:5.1f} %)" for o in out), flush=True)
        sim.close()
