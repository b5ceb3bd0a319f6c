"""Development: the 11v11 scrimmage task over batch sizes, 32-lanes-per-env kernels (RSX_LAYOUT=lanes) against the
four-lanes-per-env kernel of rsx_quad_ssl.hpp (RSX_LAYOUT=quad)."""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
NB = 2 * 4 * (5 + 11 * 22) + 4 * 8 * 22 + 4 * 46 + 5
for task, name in ((6, "spread"), (7, "crowded")):
    for B, n in ((16384, 200), (32768, 200), (65536, 100), (131072, 60), (262144, 40)):
        sim = L.Sim(1, 1, 11, 11, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
        out = []
        for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
            fn(n); torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
            out.append((time.perf_counter() - t) / n * 1e6)
        print(f"11v11 {name:8s} {B:8d} step {out[0]:8.2f} us ({NB*B/out[0]/8e4:5.1f} % of 8 TB/s)  one-launch {out[1]:8.2f} us/step ({NB*B/out[1]/8e4:5.1f} %)", flush=True)
        sim.close()
'''
for lay in ("lanes", "quad"):
    print("== RSX_LAYOUT=" + lay, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LAYOUT=lay))
