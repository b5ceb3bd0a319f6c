"""Development: separate what a model / kernel change costs in INSTRUCTIONS from what it costs through the states it leads to.
Each library build evolves VSS-v0 (4096 envs) for 4000 steps and saves a checkpoint; then every build is timed from every checkpoint
(300 steps per repetition, reloaded each time: the populations have no time to drift far from the saved one).
    python tools/exp_cross_state.py libA.so libB.so ..."""
import os, subprocess, sys
MAKE = r'''
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from rsoccer_amd import _lib as L
sim = L.Sim(0, 0, 3, 3, 25, 4096); sim.task_attach(1, 0, 0, 0); sim.task_reset()
sim.task_step_n(4000, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
np.save(sys.argv[1], sim.task_checkpoint())
'''
TIME = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
sim = L.Sim(0, 0, 3, 3, 25, 4096); sim.task_attach(1, 0, 0, 0); sim.task_reset()
out = []
for ck in sys.argv[1:]:
    blob = np.load(ck)
    ts = []
    for rep in range(12):
        sim.task_restore(blob); torch.cuda.synchronize()
        t = time.perf_counter(); sim.task_step_n(300, s); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / 300 * 1e6)
    out.append(f"{sorted(ts)[len(ts) // 2]:6.2f}")
print("  ".join(out), flush=True)
'''
libs = sys.argv[1:]
cks = []
for lib in libs:
    ck = f"/tmp/ck_{os.path.basename(lib)}.npy"
    subprocess.run([sys.executable, "-c", MAKE, ck], env=dict(os.environ, RSX_LIB=lib), check=True)
    cks.append(ck)
print("timed build \\ states evolved by:", "  ".join(os.path.basename(l) for l in libs))
for rnd in range(2):
    for lib in libs:
        r = subprocess.run([sys.executable, "-c", TIME] + cks, env=dict(os.environ, RSX_LIB=lib), capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}", flush=True)
