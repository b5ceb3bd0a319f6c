"""Occupancy sensitivity of the one-lane-per-env VSS-v0 step kernel (development): dynamic LDS padding limits the
workgroups per CU (RSX_EPL_LDS_PAD, bytes).  python tools/exp_occupancy.py"""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for B, n in ((1 << 20, 100), (1 << 22, 40)):
    sim = L.Sim(0, 0, 3, 3, 25, B); sim.task_attach(1, 0, 0, 0); sim.task_reset()
    sim.task_step_n(n, s); torch.cuda.synchronize(); t = time.perf_counter(); sim.task_step_n(n, s); torch.cuda.synchronize()
    us = (time.perf_counter() - t) / n * 1e6
    print(f"pad {os.environ.get('RSX_EPL_LDS_PAD','0'):>6s} B  vss {B:8d} step {us:8.2f} us ({541*B/us/8e6:5.1f} %)", flush=True)
    sim.close()
'''
for pad in sys.argv[1:] or ["0", "12500", "30000"]:
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_EPL_LDS_PAD=pad))
