"""In-kernel timeline of single-step launches (needs a -DRSX_TIMING build: RSX_LIB=...)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd import _lib as L
B = 4096
nb = 512
dbg = torch.zeros(8 * nb, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
L.load().rsx_dbg_set(ctypes.c_void_p(dbg.data_ptr()))   # before ANY launch: every kernel stamps
sim = L.Sim(0, 0, 3, 3, 25, B); sim.task_attach(1, 0, 0, 1 << 30); sim.task_reset()
s = torch.cuda.current_stream().cuda_stream
sim.task_step_n(500, s); torch.cuda.synchronize()
acc = []
for it in range(200):
    sim.task_step(None, s); torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(8, nb).astype(np.float64)
    acc.append(d)
d = np.stack(acc)            # [it, stamp, block]
t0 = d[:, 0].min(axis=1, keepdims=True)   # first wave start per launch
names = ["entry", "loads landed", "cmds done", "physics done", "epilogue done", "before stores", "stores issued", "stores acked"]
print("cycles are s_memtime ticks (100 MHz constant clock on this GPU?) -> reported raw")
for i, n in enumerate(names):
    rel = d[:, i] - t0
    print(f"{n:16s} mean {rel.mean():9.1f}  min-wave {rel.min(axis=1).mean():9.1f}  max-wave {rel.max(axis=1).mean():9.1f}")
print("per-wave durations: load", (d[:,1]-d[:,0]).mean(), "cmds", (d[:,2]-d[:,1]).mean(), "physics", (d[:,3]-d[:,2]).mean(),
      "epilogue", (d[:,4]-d[:,3]).mean(), "reset blk", (d[:,5]-d[:,4]).mean(), "stores", (d[:,6]-d[:,5]).mean(), "ack", (d[:,7]-d[:,6]).mean())
