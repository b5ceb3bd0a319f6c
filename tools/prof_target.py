"""Profiling target: a few launches of the fused VSS-v0 step in each mode (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "step"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
sim = L.Sim(0, 0, 3, 3, 25, B)
sim.task_attach(1, 0, 0, 0)
sim.task_reset()
torch.cuda.synchronize()
if mode == "step":
    for _ in range(n):
        sim.task_step(None)
elif mode == "rollout":
    sim.task_rollout(n)
torch.cuda.synchronize()
print("done", sim.read_metrics())
