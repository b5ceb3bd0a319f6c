"""Development: per-step time of the large-batch legs by row pad (RSX_ROW_PAD, floats), each pad in its own process.
python tools/exp_row_pad.py [pad ...]    (profiles/r05_row_stride.txt)"""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
out = []
for name, kind, ft, nb, ny, task, B, n in (("vss1M", 0, 0, 3, 3, 1, 1 << 20, 100), ("sd1M", 1, 2, 1, 6, 2, 1 << 20, 100), ("vss2M", 0, 0, 3, 3, 1, 1 << 21, 60), ("vss4M", 0, 0, 3, 3, 1, 1 << 22, 40), ("sd4M", 1, 2, 1, 6, 2, 1 << 22, 40)):
    best = []
    for rep in range(2):
        sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
        sim.task_step_n(200, s); torch.cuda.synchronize()
        t = time.perf_counter(); sim.task_step_n(n, s); torch.cuda.synchronize(); best.append((time.perf_counter() - t) / n * 1e6)
        sim.close()
    out.append(f"{name} {best[0]:6.1f} {best[1]:6.1f}")
print("  ".join(out), flush=True)
'''
pads = sys.argv[1:] or ["0", "4160", "8256", "16448", "16512", "24640", "32832", "49216", "65600", "98368", "131136"]
for pad in pads:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_ROW_PAD=pad), capture_output=True, text=True)
    print(f"pad {pad:>7s}  {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
