"""Development: every registered task over the mid-range batch sizes, lane-group kernels against the one-lane-per-env kernels
(RSX_LAYOUT=lanes / epl) - where rsx_api.hip switches layouts.  python tools/layout_crossovers.py"""
import sys, os, time, subprocess
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for name, kind, ft, nb, ny, task in (("vss", 0, 0, 3, 3, 1), ("sd", 1, 2, 1, 6, 2), ("drib", 1, 2, 1, 4, 3), ("cont", 1, 2, 1, 1, 4), ("pass", 1, 2, 2, 0, 5)):
    for B in (16384, 32768, 49152, 65536, 98304, 131072):
        sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
        n = 300
        sim.task_step_n(n, s); torch.cuda.synchronize(); t = time.perf_counter(); sim.task_step_n(n, s); torch.cuda.synchronize()
        us = (time.perf_counter() - t) / n * 1e6
        print(f"{os.environ['RSX_LAYOUT']:5s} {name:5s} {B:8d} step {us:8.2f} us", flush=True)
        sim.close()
'''
for lay in ("lanes", "epl"):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LAYOUT=lay))
