"""In-kernel timeline of the SSL one-lane-per-env kernel (needs tools/build_timing.sh; RSX_LIB=tools/_dev/librsx_hip_timing.so).
usage: B=65536 TASK=6 python tools/exp_timeline_epl.py"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
os.environ["RSX_LAYOUT"] = "epl"
from rsoccer_amd import _lib as L
B = int(os.environ.get("B", 65536)); task = int(os.environ.get("TASK", 6))
kind, ft, nb_, ny = {2: (1, 2, 1, 6), 3: (1, 2, 1, 4), 4: (1, 2, 1, 1), 5: (1, 2, 2, 0), 6: (1, 1, 11, 11), 7: (1, 1, 11, 11)}[task]
NS = 20
nblk = ((B + 63) // 64 + 7) // 8 * 8
dbg = torch.zeros(NS * B, dtype=torch.int64, device="cuda")   # (reset() runs the lanes kernel: up to B / 2 blocks stamp too)
torch.cuda.synchronize()
L.load().rsx_dbg_set(ctypes.c_void_p(dbg.data_ptr()))
sim = L.Sim(kind, ft, nb_, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
s = torch.cuda.current_stream().cuda_stream
sim.task_step_n(50, s); torch.cuda.synchronize()
acc = []
for it in range(30):
    sim.task_step(None, s); torch.cuda.synchronize()
    acc.append(dbg[: NS * nblk].cpu().numpy().reshape(NS, nblk).astype(np.float64))
d = np.stack(acc)[:, :, : (B + 63) // 64]
seq = [("loads", 0, 1), ("interpret (sincos)", 1, 2), ("commands (philox, targets)", 2, 3), ("sub0: actuation + integration", 3, 4),
       ("sub0: pair tests + near tests", 4, 5), ("sub0: contact sweeps", 5, 6), ("sub0: walls", 6, 7), ("whole physics (5 sub-steps)", 3, 8),
       ("wire format + obs + robot stores", 8, 9), ("reward + flags", 9, 10), ("episode end + obs store", 10, 11), ("final stores issued", 11, 12),
       ("stores acked", 12, 13), ("whole wave", 0, 13)]
for n, a, b in seq:
    x = d[:, b] - d[:, a]
    print(f"  {n:36s} mean {x.mean():9.0f} cycles  median {np.median(x):9.0f}  p95 {np.percentile(x, 95):9.0f}")
