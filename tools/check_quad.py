"""Development: the four-lanes-per-env kernel of the SSL 11v11 scrimmage task (RSX_LAYOUT=quad) against the 32-lane
kernel, bit for bit, then its speed at 65 536 envs.  python tools/check_quad.py [lib ...]"""
import os, subprocess, sys
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
def run(layout, task, B, steps, max_steps):
    os.environ["RSX_LAYOUT"] = layout
    sim = L.Sim(1, 1, 11, 11, 25, B); sim.task_attach(task, 31337, 77, max_steps); tens = sim.task_tensors(); sim.task_reset()
    rng = np.random.default_rng(5)
    snaps = []
    for t in range(steps):
        if t % 3 == 0:
            a = rng.uniform(-1, 1, tuple(tens["actions"].shape)).astype(np.float32)
            tens["actions"].copy_(torch.from_numpy(a)); sim.task_step(tens["actions"].data_ptr())
        else:
            sim.task_step(None)
        if t % 25 == 24 or t == steps - 1:
            torch.cuda.synchronize()
            snaps.append(np.concatenate([sim.get_state_full().ravel()] + [tens[k].cpu().numpy().astype(np.float64).ravel()
                         for k in ("obs", "reward", "terminated", "truncated", "info", "final_obs", "steps")] + [sim.read_metrics().astype(np.float64)]))
    sim.close()
    return snaps
for task in (7, 6):
    for B, steps, ms in ((150, 200, 45), (333, 400, 0)):
        a = run("lanes", task, B, steps, ms); b = run("quad", task, B, steps, ms)
        ok = True
        for i, (x, y) in enumerate(zip(a, b)):
            if not np.array_equal(x, y, equal_nan=True):
                ok = False
                d = np.flatnonzero(~((x == y) | (np.isnan(x) & np.isnan(y))))
                S = B * 249
                print(f"  MISMATCH task {task} B={B} snapshot {i}: {d.size} values differ; in state: {np.count_nonzero(d < S)}, first at {d[:6]} (state cols {[(int(v) % 249) for v in d[:6] if v < S]})", flush=True)
                break
        print(f"task {task} B={B} steps={steps} max_steps={ms}: {'bit-identical' if ok else 'DIFFERENT'}", flush=True)
for lay in ("lanes", "quad"):
    os.environ["RSX_LAYOUT"] = lay
    for task in (6, 7):
        B, n = 65536, 60
        sim = L.Sim(1, 1, 11, 11, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
        sim.task_step_n(n, s); torch.cuda.synchronize(); t = time.perf_counter(); sim.task_step_n(n, s); torch.cuda.synchronize()
        us = (time.perf_counter() - t) / n * 1e6
        print(f"{lay:5s} task {task} {B:8d} step {us:8.2f} us ({2869*B/us/8e6*100:5.1f} % of 8 TB/s)", flush=True)
        sim.close()
'''
for lib in sys.argv[1:] or ["rsoccer_amd/librsx_hip.so"]:
    print("==", lib, flush=True)
    subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, RSX_LIB=os.path.abspath(lib)))
