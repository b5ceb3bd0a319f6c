import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
# correctness vs the default library is checked by comparing states after 150 steps (file)
sim = L.Sim(0, 0, 3, 3, 25, 100); sim.task_attach(1, 3, 0, 40); sim.task_reset(); sim.task_step_n(150, s); torch.cuda.synchronize()
np.save(sys.argv[1], sim.get_state_full()); sim.close()
for B in (4096, 36864, 1048576):
    sim = L.Sim(0, 0, 3, 3, 25, B); sim.task_attach(1, 0, 0, 0); sim.task_reset()
    K = 2000 if B < 100000 else 200
    sim.task_step_n(K // 5, s); torch.cuda.synchronize()
    t = time.perf_counter(); sim.task_step_n(K, s); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
    print("PACK7" if os.environ.get("RSX_PACK7") else "L=8  ", B, round(dt * 1e6, 2), "us/step", round(B / dt / 1e6, 1), "M env-steps/s")
    sim.close()
