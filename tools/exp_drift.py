import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
sim = L.Sim(0, 0, 3, 3, 25, 4096); sim.task_attach(1, 0, 0, 0); sim.task_reset()
s = torch.cuda.current_stream().cuda_stream
sim.task_step_n(200, s); torch.cuda.synchronize()
out = []
for c in range(24):
    t = time.perf_counter(); sim.task_step_n(1000, s); torch.cuda.synchronize()
    out.append((time.perf_counter() - t) / 1000 * 1e6)
print(" ".join(f"{x:.2f}" for x in out))
print(sim.read_metrics())
