"""The raw VSS 3v3 step with 8 lanes per env against the four-lanes-per-env experiment (rsx_quad_vss.hpp): device-drawn
commands, steady state.  RSX_LAYOUT=lanes|quad python tools/quad_vss_ab.py [n_timed [batch ...]]   (n_timed = 0: a few
launches only, for rocprofv3 --pmc)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
batches = [int(a) for a in sys.argv[2:]] or [4096, 16384, 65536]
s = torch.cuda.current_stream().cuda_stream
out = []
for B in batches:
    sim = L.Sim(0, 0, 3, 3, 25, B)
    sim.step_dev_random(2000 if n else 300, 1, 0, s); torch.cuda.synchronize()
    if n == 0:
        sim.step_dev_random(20, 1, 2000, s); torch.cuda.synchronize()
        out.append(f"{B}: profiled")
    else:
        ts = []
        for rep in range(3):
            t = time.perf_counter(); sim.step_dev_random(n, 1, 2000 + n * rep, s); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / n * 1e6)
        out.append(f"{B}: {min(ts):.3f} us (state checksum {float(np.abs(sim.get_state_full()).sum()):.4f})")
    sim.close()
print("RSX_LAYOUT=%s raw VSS 3v3 step, device-drawn commands: " % os.environ.get("RSX_LAYOUT", "default") + "   ".join(out), flush=True)
