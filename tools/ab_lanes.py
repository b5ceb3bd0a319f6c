"""A/B of the lanes-per-env mapping at the BASELINE batch sizes (development tool):
python tools/ab_lanes.py  -> VSS-v0 3v3 @ 4096 and SSLStaticDefenders 1v6 @ 2048 with 8 and 16 lanes per env,
per-step launches and one launch, plus a bit-comparison of the two mappings after 300 steps."""
import os, subprocess, sys
CHILD = r'''
import sys, os, time, hashlib
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for name, kind, ft, nb, ny, task, B in (("vss", 0, 0, 3, 3, 1, 4096), ("sd", 1, 2, 1, 6, 2, 2048), ("vss", 0, 0, 3, 3, 1, 16384), ("vss", 0, 0, 3, 3, 1, 65536)):
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 7, 0, 0); sim.task_reset()
    sim.task_step_n(300, s); torch.cuda.synchronize()
    v = sim.task_tensors()
    h = hashlib.sha1(v["obs"].cpu().numpy().tobytes()).hexdigest()[:12]
    out = []
    for fn in (lambda k: sim.task_step_n(k, s), lambda k: sim.task_rollout(k, s)):
        fn(2000); torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            t = time.perf_counter(); fn(4000); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t) / 4000 * 1e6)
        out.append(best)
    print(f"{name:4s} {B:6d} L={os.environ.get('RSX_LANES_PER_ENV','8'):2s} step {out[0]:7.2f} us  one-launch {out[1]:7.2f} us/step  obs@300 {h}", flush=True)
    sim.close()
'''
for lanes in ("8", "16"):
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LANES_PER_ENV=lanes, RSX_LAYOUT="lanes"))
