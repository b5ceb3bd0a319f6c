"""Development: memory-only launches of the one-lane-per-env kernels (rsx_task_rollout(0): load every state row, store it
back, no step) against full steps, 1 M and 4 M envs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for name, kind, ft, nb, ny, task, rows in (("vss", 0, 0, 3, 3, 1, 43), ("sd", 1, 2, 1, 6, 2, 84)):
    for B in (1 << 20, 1 << 22):
        if name == "sd" and B > (1 << 20):
            continue
        sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
        out = []
        for fn in (lambda k: [sim.task_rollout(0, s) for _ in range(k)], lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
            fn(30); torch.cuda.synchronize(); t = time.perf_counter(); fn(30); torch.cuda.synchronize()
            out.append((time.perf_counter() - t) / 30 * 1e6)
        moved = 2 * rows * 4 * B
        print(f"{name} {B:8d}: load+store only {out[0]:8.2f} us ({moved / out[0] / 1e6:6.2f} TB/s of state traffic)   step {out[1]:8.2f} us   one-launch {out[2]:8.2f} us/step", flush=True)
        sim.close()
