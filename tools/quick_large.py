"""Development: the large-batch legs only (one-lane-per-env kernels), per-step launches and one launch.
python tools/quick_large.py [lib ...]"""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
BYTES = {1: 541, 2: 981, 3: 729, 4: 341, 5: 349}
for name, kind, ft, nb, ny, task, B, n in (("vss", 0, 0, 3, 3, 1, 1 << 20, 100), ("vss", 0, 0, 3, 3, 1, 1 << 22, 40), ("sd", 1, 2, 1, 6, 2, 1 << 20, 60),
        ("drib", 1, 2, 1, 4, 3, 1 << 20, 60), ("cont", 1, 2, 1, 1, 4, 1 << 20, 60), ("pass", 1, 2, 2, 0, 5, 1 << 20, 60)):
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
    out = []
    for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
        fn(n); torch.cuda.synchronize()
        best = 1e9
        for rep in range(2):
            t = time.perf_counter(); fn(n); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / n * 1e6)
        out.append(best)
    print(f"{name:5s} {B:8d} step {out[0]:8.2f} us ({BYTES[task]*B/out[0]/8e6*100:5.1f} %)  one-launch {out[1]:8.2f} us/step", flush=True)
    sim.close()
'''
for lib in sys.argv[1:] or ["rsoccer_amd/librsx_hip.so"]:
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LIB=os.path.abspath(lib)))
