"""Development: VSS-v0 large-batch legs only (one-lane-per-env kernel).  python tools/quick_epl.py [lib ...]"""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for B, n in ((1 << 20, 100), (1 << 22, 60)):
    sim = L.Sim(0, 0, 3, 3, 25, B); sim.task_attach(1, 0, 0, 0); sim.task_reset()
    out = []
    for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
        fn(n); torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / n * 1e6)
    print(f"vss {B:8d} step {out[0]:8.2f} us ({541*B/out[0]/8e6:5.1f} % of 8 TB/s)  one-launch {out[1]:8.2f} us/step ({541*B/out[1]/8e6:5.1f} %)", flush=True)
    sim.close()
'''
for lib in sys.argv[1:] or ["rsoccer_amd/librsx_hip.so"]:
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LIB=os.path.abspath(lib)))
