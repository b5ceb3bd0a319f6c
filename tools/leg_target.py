"""Profiling target for ONE large-batch leg of bench.py (tools/prof_leg_traffic.sh runs it under rocprofv3 --pmc):
python tools/leg_target.py <leg>   with <leg> = <task>:<envs>, task in vss | sd | drib | cont | pass | scrim | scrimC.
Warm-up steps run inside ONE launch (another kernel name) where the handle offers it, then N per-step launches."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L

TASKS = {"vss": (0, 0, 3, 3, 1), "sd": (1, 2, 1, 6, 2), "drib": (1, 2, 1, 4, 3), "cont": (1, 2, 1, 1, 4),
         "pass": (1, 2, 2, 0, 5), "scrim": (1, 1, 11, 11, 6), "scrimC": (1, 1, 11, 11, 7)}
name, envs = sys.argv[1].split(":")
envs = int(envs)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
warm = int(sys.argv[3]) if len(sys.argv) > 3 else (30 if name.startswith("scrim") else 100)
kind, ft, nb, ny, task = TASKS[name]
sim = L.Sim(kind, ft, nb, ny, 25, envs)
sim.task_attach(task, 0, 0, 0)
sim.task_reset()
sim.task_rollout(warm)          # warm-up: steady-state contacts / episode phases before the profiled launches
torch.cuda.synchronize()
for _ in range(n):
    sim.task_step(None)
torch.cuda.synchronize()
print("leg", sys.argv[1], "layout", sim.task_layout(), "profiled launches", n, flush=True)
sim.close()
