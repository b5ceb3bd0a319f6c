"""Development: the SSL tasks, 8-lanes-per-env vs one-lane-per-env kernel over batch sizes.
usage: python tools/quick_ssl.py [task ids, default 2 3 4 5]"""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
CFG = {2: ("static-defenders", 1, 6, 981), 3: ("dribbling", 1, 4, 2 * 4 * 60 + 4 * 40 + 4 * 21 + 5),
       4: ("contested", 1, 1, 2 * 4 * 27 + 4 * 16 + 4 * 14 + 5), 5: ("pass-endurance", 2, 0, 2 * 4 * 27 + 4 * 16 + 4 * 16 + 5)}
for task in [int(a) for a in sys.argv[1:]]:
    name, nb, ny, nbytes = CFG[task]
    for B, n in ((16384, 300), (65536, 200), (131072, 200), (262144, 100), (1 << 20, 60)):
        sim = L.Sim(1, 2, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
        out = []
        for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
            fn(n); torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
            out.append((time.perf_counter() - t) / n * 1e6)
        print(f"{name:16s} {B:8d} step {out[0]:8.2f} us ({nbytes*B/out[0]/8e4:5.1f} % of 8 TB/s)  one-launch {out[1]:8.2f} us/step ({nbytes*B/out[1]/8e4:5.1f} %)", flush=True)
        sim.close()
'''
tasks = sys.argv[1:] or ["2", "3", "4", "5"]
for lay in ("lanes", "epl"):
    print("== RSX_LAYOUT=" + lay, flush=True)
    subprocess.run([sys.executable, "-c", CHILD] + tasks, env=dict(os.environ, RSX_LAYOUT=lay))
