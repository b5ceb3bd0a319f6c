#!/bin/bash
# A/B of two builds over the fused task configs (development tool): tools/ab_configs.sh libA libB
for lib in "$@"; do echo "== $lib"; RSX_LIB=$lib python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream
for name, kind, ft, nb, ny, task, B in (("vss", 0, 0, 3, 3, 1, 4096), ("sd", 1, 2, 1, 6, 2, 2048), ("drib", 1, 2, 1, 4, 3, 2048), ("cont", 1, 2, 1, 1, 4, 2048), ("pass", 1, 2, 2, 0, 5, 2048)):
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
    out = []
    for fn in (lambda n: sim.task_step_n(n, s), lambda n: sim.task_rollout(n, s)):
        fn(300); torch.cuda.synchronize(); t = time.perf_counter(); fn(2000); torch.cuda.synchronize()
        out.append((time.perf_counter() - t) / 2000 * 1e6)
    print(f"{name:5s} step {out[0]:6.2f} us  one-launch {out[1]:6.2f} us/step", flush=True)
    sim.close()
PY
done
