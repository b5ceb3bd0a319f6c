"""Development: is the power-of-two row stride (4 M envs x 4 B = 16 MB) hurting?  Same kernel, batch sizes around 2^22."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for B in (4194304, 4194304 - 4096, 4194304 + 4096, 4000000, 4194304, 1048576, 1048576 + 1024, 1000000):
    sim = L.Sim(0, 0, 3, 3, 25, B); sim.task_attach(1, 0, 0, 0); sim.task_reset()
    out = []
    for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
        fn(40); torch.cuda.synchronize(); t = time.perf_counter(); fn(40); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / 40 * 1e6)
    print(f"B {B:8d}: step {out[0]:8.2f} us = {out[0] / B * 1e6:7.2f} ps/env ({541*B/out[0]/8e4:5.1f} %)   one-launch {out[1]:8.2f} us = {out[1] / B * 1e6:7.2f} ps/env", flush=True)
    sim.close()
