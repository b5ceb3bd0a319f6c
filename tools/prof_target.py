"""Profiling target: a few launches of the fused VSS-v0 step in each mode (for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "step"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 200
# optional 4th argument: task id (1 VSS-v0 3v3, 2 static defenders, 3 dribbling, 4 contested, 5 pass endurance, 6/7 11v11 scrimmage)
task = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kind, ft, nb, ny = {1: (0, 0, 3, 3), 2: (1, 2, 1, 6), 3: (1, 2, 1, 4), 4: (1, 2, 1, 1), 5: (1, 2, 2, 0), 6: (1, 1, 11, 11), 7: (1, 1, 11, 11)}[task]
sim = L.Sim(kind, ft, nb, ny, 25, B)
sim.task_attach(task, 0, 0, 0)
sim.task_reset()
# optional 5th argument: steps of warm-up in ONE launch first (another kernel name), so that the profiled launches are steady-state steps
if len(sys.argv) > 5 and int(sys.argv[5]) > 0:
    sim.task_rollout(int(sys.argv[5]))
torch.cuda.synchronize()
if mode == "step":
    for _ in range(n):
        sim.task_step(None)
elif mode == "rollout":
    sim.task_rollout(n)
torch.cuda.synchronize()
print("done", sim.read_metrics())
