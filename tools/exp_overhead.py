"""Where does a single-step launch spend its time?  Times K back-to-back launches of the fused
VSS-v0 kernel with 0, 1, 2, 4, 8 steps per launch (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
sim = L.Sim(0, 0, 3, 3, 25, B)
sim.task_attach(1, 0, 0, 0)
sim.task_reset()
s = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
K = 2000
res = {}
for n in (0, 1, 2, 4, 8, 16):
    for _ in range(200):
        sim.task_rollout(n, s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(K):
        sim.task_rollout(n, s)
    torch.cuda.synchronize()
    res[n] = (time.perf_counter() - t) / K * 1e6
print("B", B, "HIP_FORCE_DEV_KERNARG", os.environ.get("HIP_FORCE_DEV_KERNARG"), {k: round(v, 2) for k, v in res.items()},
      "per-step", round((res[16] - res[8]) / 8, 2), "floor(0 steps)", round(res[0], 2))
