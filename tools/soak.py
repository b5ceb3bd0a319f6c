"""Long soak on the GPU (development tool): millions of steps per task, invariants checked at the
end of every chunk (finite state, bodies inside the walls, counters consistent)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd import _lib as L
SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0   # fraction of the full soak
CASES = [("vss", 0, 0, 3, 3, 1, 4096, 1_000_000), ("vss-epl", 0, 0, 3, 3, 1, 131072, 60_000), ("sd", 1, 2, 1, 6, 2, 2048, 400_000),
         ("sd-epl", 1, 2, 1, 6, 2, 131072, 40_000), ("drib", 1, 2, 1, 4, 3, 2048, 300_000), ("cont", 1, 2, 1, 1, 4, 2048, 300_000),
         ("pass", 1, 2, 2, 0, 5, 2048, 300_000), ("drib-epl", 1, 2, 1, 4, 3, 131072, 40_000), ("cont-epl", 1, 2, 1, 1, 4, 131072, 40_000),
         ("pass-epl", 1, 2, 2, 0, 5, 131072, 40_000), ("vss5v5", 0, 1, 5, 5, 1, 1024, 200_000),
         ("scrim", 1, 1, 11, 11, 6, 1024, 100_000), ("scrim-crowded", 1, 1, 11, 11, 7, 1024, 100_000),
         ("scrim-quad", 1, 1, 11, 11, 6, 65536, 20_000), ("scrim-crowded-quad", 1, 1, 11, 11, 7, 131072, 6_000),
         ("vss-epl-1M", 0, 0, 3, 3, 1, 1 << 20, 8_000), ("sd-epl-1M", 1, 2, 1, 6, 2, 1 << 20, 6_000)]
CASES = [c[:7] + (max(1000, int(c[7] * SCALE)),) for c in CASES]
for name, kind, ft, nb, ny, task, B, steps in CASES:
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 7, 0, 0); sim.task_reset()
    tens = sim.task_tensors()
    f = sim.get_field_params()
    lim_x = f["length"] / 2 + f["goal_depth"] + (0.0 if kind == 0 else 0.35) + 1e-4
    lim_y = f["width"] / 2 + (0.0 if kind == 0 else 0.35) + 1e-4
    t0 = time.time(); done = 0
    while done < steps:
        n = min(50_000, steps - done)
        sim.task_step_n(n - n // 4)
        sim.task_rollout(n // 4)
        done += n
        torch.cuda.synchronize()
        st = sim.state_tensor()
        assert torch.isfinite(st).all(), name
        rs = 6 if kind == 0 else 11
        xs = torch.stack([st[0]] + [st[5 + rs * k] for k in range(nb + ny)])
        ys = torch.stack([st[1]] + [st[6 + rs * k] for k in range(nb + ny)])
        assert xs.abs().max().item() <= lim_x and ys.abs().max().item() <= lim_y, (name, xs.abs().max().item(), ys.abs().max().item())
        assert torch.isfinite(tens["obs"]).all() and torch.isfinite(tens["reward"]).all()
        assert sim.check_finite() == 0
    m = sim.read_metrics()
    assert m[0] == B * steps and m[5] <= m[0]
    print(f"{name:18s} {B:7d} envs x {steps:8d} steps ok: {B * steps / 1e9:6.2f} G env-steps in {time.time() - t0:5.1f} s, episodes {m[1]}, truncated {m[6]}", flush=True)
    sim.close()
