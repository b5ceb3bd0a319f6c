"""Development: SSLStaticDefenders, 8-lanes-per-env vs one-lane-per-env kernel over batch sizes."""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for B, n in ((16384, 300), (32768, 300), (65536, 200), (131072, 200), (262144, 100), (1 << 20, 60)):
    sim = L.Sim(1, 2, 1, 6, 25, B); sim.task_attach(2, 0, 0, 0); sim.task_reset()
    out = []
    for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
        fn(n); torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / n * 1e6)
    print(f"sd {B:8d} step {out[0]:8.2f} us ({981*B/out[0]/8e4:5.1f} % of 8 TB/s)  one-launch {out[1]:8.2f} us/step ({981*B/out[1]/8e4:5.1f} %)", flush=True)
    sim.close()
'''
for lay in ("lanes", "epl"):
    print("== RSX_LAYOUT=" + lay, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LAYOUT=lay))
