"""How much of a single-step launch is the reset tail?  Compares the default TimeLimit (resets
spread over the batch every step) with an effectively infinite one (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for max_steps in (0, 1 << 30):
    sim = L.Sim(0, 0, 3, 3, 25, 4096)
    sim.task_attach(1, 0, 0, max_steps)
    sim.task_reset()
    sim.task_step_n(3000, s)           # decorrelate episode phases
    torch.cuda.synchronize()
    m0 = sim.read_metrics()
    t = time.perf_counter(); sim.task_step_n(3000, s); torch.cuda.synchronize(); dt = time.perf_counter() - t
    m1 = sim.read_metrics()
    print("max_steps", max_steps, "us/step", round(dt / 3000 * 1e6, 2), "episodes ended per step", (m1[1] - m0[1]) / 3000)
    sim.close()
