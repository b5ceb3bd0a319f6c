"""Cost of the in-step auto-reset path: step time with TimeLimit = 1 (every env resets in every
step) vs no episode end at all (development tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for ms in (0, 1, 2, 8, 64):
    sim = L.Sim(0, 0, 3, 3, 25, B)
    sim.task_attach(1, 0, 0, ms)
    sim.task_reset()
    s = torch.cuda.current_stream().cuda_stream
    sim.task_step_n(300, s); torch.cuda.synchronize()
    K = 2000
    t = time.perf_counter(); sim.task_step_n(K, s); torch.cuda.synchronize(); dt = time.perf_counter() - t
    m = sim.read_metrics()
    print(f"max_episode_steps={ms:3d}: {dt / K * 1e6:7.2f} us/step, resets per step {m[1] / (K + 300):8.1f}", flush=True)
    sim.close()
