"""Development: where the cycles of a sub-step go in the lane-group kernels (A: actuation + integration, B: contacts incl. the
LDS exchange, C: walls + the closing wave_sync), sub-steps 1.. of single-step launches.  Needs a -DRSX_TIMING -DRSX_TIMING_SUB
build (tools/build_variant.sh subt -DRSX_TIMING -DRSX_TIMING_SUB; RSX_LIB=tools/_dev/librsx_subt.so); CFG / B / LANES as in
exp_timeline2.py.  The stamps (s_memtime + a wait on lgkmcnt) add ~5 % to the sub-step themselves."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd import _lib as L
B = int(os.environ.get("B", 4096)); NS = 20
G = 64 // int(os.environ.get("LANES", 8))
nb = ((B + G - 1) // G + 7) // 8 * 8
cfg = os.environ.get("CFG", "vss")
HELP = (B + 63) // 64 if cfg == "sd" and not os.environ.get("RSX_NO_PCACHE") and B <= 16384 else 0
dbg = torch.zeros(NS * (nb + HELP), dtype=torch.int64, device="cuda"); torch.cuda.synchronize()
L.load().rsx_dbg_set(ctypes.c_void_p(dbg.data_ptr()))
CFG = {"vss": (0, 0, 3, 3, 1), "sd": (1, 2, 1, 6, 2), "drib": (1, 2, 1, 4, 3), "scrim": (1, 1, 11, 11, 6)}[cfg]
sim = L.Sim(CFG[0], CFG[1], CFG[2], CFG[3], 25, B); sim.task_attach(CFG[4], 0, 0, 0); sim.task_reset()
s = torch.cuda.current_stream().cuda_stream
sim.task_step_n(500, s); torch.cuda.synchronize()
acc = []
for it in range(100):
    sim.task_step(None, s); torch.cuda.synchronize()
    acc.append(dbg.cpu().numpy()[:NS * (nb + HELP)].reshape(NS, nb + HELP)[:, :nb].astype(np.float64))
d = np.stack(acc)
used = d[0, 0] > 0
d = d[:, :, used]
n = 4.0
for name, k in (("A actuation + integration", 15), ("B contacts (publish, pair test, walks)", 16), ("C walls + closing sync", 17)):
    x = d[:, k] / n
    print(f"{name:42s} per sub-step: mean {x.mean():7.0f}  median {np.median(x):7.0f}  p95 {np.percentile(x, 95):7.0f}  p99 {np.percentile(x, 99):7.0f} cycles")
x = (d[:, 12] - d[:, 8]) / n
print(f"{'whole sub-step (1..4)':42s} per sub-step: mean {x.mean():7.0f}  median {np.median(x):7.0f}  p95 {np.percentile(x, 95):7.0f}  p99 {np.percentile(x, 99):7.0f} cycles")
