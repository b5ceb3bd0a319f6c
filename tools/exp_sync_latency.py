"""Development: what the closing synchronisation of a SHORT timed region costs (the driver's --steps 20 region is 20 x 9 us of
GPU work + one launch latency + one wake-up), under the runtime's wait policies.  python tools/exp_sync_latency.py"""
import os, subprocess, sys
CHILD = r'''
import sys, os, time, statistics
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
sim = L.Sim(0, 0, 3, 3, 25, 4096); sim.task_attach(1, 0, 0, 0); sim.task_reset()
sim.task_step_n(300, s); torch.cuda.synchronize()
res = {}
for n in (1, 20, 200, 2000):
    ts = []
    for _ in range(15):
        sim.task_step_n(5, s); torch.cuda.synchronize()
        t = time.perf_counter(); sim.task_step_n(n, s); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e6)
    res[n] = statistics.median(ts)
ts = []
for _ in range(15):   # the bench's bracket: two timing events inside the region
    sim.task_step_n(5, s); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record(); sim.task_step_n(20, s); e1.record(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e6)
res["20+events"] = statistics.median(ts)
print("  ".join(f"n={n}: {v:8.1f} us" + (f" ({v / n:5.2f})" if isinstance(n, int) else "") for n, v in res.items()), flush=True)
'''
for tag, env in (("default", {}), ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}), ("ROC_ACTIVE_WAIT_TIMEOUT=1000", {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"}),
                 ("both", {"HSA_ENABLE_INTERRUPT": "0", "ROC_ACTIVE_WAIT_TIMEOUT": "1000"}), ("default again", {})):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    print(f"{tag:30s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-400:]}", flush=True)
