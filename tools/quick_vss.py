"""Development: per-step launch time of VSS-v0 at 4096 envs (the headline leg) for a list of library builds, each in its own process,
interleaved rounds (same box):  python tools/quick_vss.py [rounds] libA.so libB.so ..."""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
sim = L.Sim(0, 0, 3, 3, 25, 4096); sim.task_attach(1, 0, 0, 0); sim.task_reset()
sim.task_step_n(2000, s); torch.cuda.synchronize()
ts = []
for _ in range(5):
    t = time.perf_counter(); sim.task_step_n(4000, s); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 4000 * 1e6)
print("vss4096 " + " ".join(f"{x:5.2f}" for x in ts) + f"  median {sorted(ts)[2]:5.2f}", flush=True)
'''
rounds = int(sys.argv[1]) if sys.argv[1].isdigit() else 2
libs = sys.argv[2:] if sys.argv[1].isdigit() else sys.argv[1:]
for rnd in range(rounds):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LIB=lib), capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
