"""In-kernel timeline incl. per-sub-step stamps, single-step launch vs the last step of a
multi-step launch (needs the -DRSX_TIMING build of tools/build_timing.sh: RSX_LIB=tools/_dev/librsx_hip_timing.so)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd import _lib as L
B = int(os.environ.get("B", 4096))
NS = 20
G = 64 // int(os.environ.get("LANES", 8))   # envs per wave (LANES=32 for the 11v11 task)
nb = ((B + G - 1) // G + 7) // 8 * 8
# SSLStaticDefenders single-step launches with the placement cache on carry ceil(B / 64) helper workgroups behind the tiles: the
# stamp rows are gridDim.x apart
HELP = (B + 63) // 64 if os.environ.get("CFG", "vss") == "sd" and not os.environ.get("RSX_NO_PCACHE") and B <= 16384 else 0
dbg = torch.zeros(NS * (nb + HELP), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
L.load().rsx_dbg_set(ctypes.c_void_p(dbg.data_ptr()))
CFG = {"vss": (0, 0, 3, 3, 1), "sd": (1, 2, 1, 6, 2), "drib": (1, 2, 1, 4, 3), "cont": (1, 2, 1, 1, 4), "pass": (1, 2, 2, 0, 5),
       "scrim": (1, 1, 11, 11, 6), "scrimC": (1, 1, 11, 11, 7)}[os.environ.get("CFG", "vss")]
sim = L.Sim(CFG[0], CFG[1], CFG[2], CFG[3], 25, B); sim.task_attach(CFG[4], 0, 0, 0); sim.task_reset()
s = torch.cuda.current_stream().cuda_stream
sim.task_step_n(500, s); torch.cuda.synchronize()
names = ["entry", "loads landed", "cmds done", "physics done", "epilogue done", "before stores", "stores issued", "stores acked",
         "sub0", "sub1", "sub2", "sub3", "sub4"]
def collect(fn, n, stride):
    acc = []
    for it in range(n):
        fn(); torch.cuda.synchronize()
        acc.append(dbg.cpu().numpy()[:NS * stride].reshape(NS, stride)[:, :nb].astype(np.float64))
    return np.stack(acc)
for label, fn in (("single-step launch", lambda: sim.task_step(None, s)),
                  ("single-step launch, last of 50 back-to-back", lambda: sim.task_step_n(50, s)),
                  ("last step of a 6-step launch", lambda: sim.task_rollout(6, s))):
    d = collect(fn, 100, nb + (HELP if label.startswith("single") else 0))
    print("==", label)
    seq = [("cmds (philox, OU, targets)", 1, 2), ("sub0", 2, 8), ("sub1", 8, 9), ("sub2", 9, 10), ("sub3", 10, 11), ("sub4", 11, 12),
           ("post-physics: obs+reward+flags", 3, 4), ("episode end + obs copy", 4, 5)]
    if label.startswith("single"):
        seq = [("loads", 0, 1)] + seq + [("stores issued", 5, 6), ("stores acked", 6, 7), ("whole wave", 0, 7)]
    for n, a, b in seq:
        x = d[:, b] - d[:, a]
        print(f"  {n:34s} mean {x.mean():8.0f}  median {np.median(x):8.0f}  p95 {np.percentile(x, 95):8.0f}  p99 {np.percentile(x, 99):8.0f}  max(avg over launches) {x.max(axis=1).mean():8.0f}")
    if label.startswith("single"):
        # the wave that finishes last decides the launch: where did it spend its time?
        used = d[0, 0] > 0
        d = d[:, :, used].copy()
        it = np.arange(d.shape[0])
        # chip-wide 100 MHz clock (s_memrealtime): when do waves start and end within a launch?
        t0 = d[:, 13].min(axis=1)
        st = np.sort(d[:, 13] - t0[:, None], axis=1).mean(axis=0) * 10.0
        en = np.sort(d[:, 14] - t0[:, None], axis=1).mean(axis=0) * 10.0
        q = [i for i in (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 384, 448, 496) if i < len(st) - 1] + [len(st) - 1]
        print("  wave start offsets ns (sorted, avg): ", [int(st[i]) for i in q])
        print("  wave end offsets ns   (sorted, avg): ", [int(en[i]) for i in q])
        last = d[:, 14].argmax(axis=1)
        print("  the last-finishing wave started at rank %.0f of %d, %.0f ns after the first" % (
            np.mean([(d[i, 13] < d[i, 13, last[i]]).sum() for i in it]), d.shape[2], np.mean(d[it, 13, last] - t0) * 10))
        rs = d[:, 18] > d[:, 4]     # waves that went through an episode end in this launch
        if rs.any():
            for n_, a_, b_ in (("reset: terminal obs + metrics", 4, 15), ("reset: predraw", 15, 16), ("reset: placement", 16, 17), ("reset: new obs", 17, 18), ("obs copy-out after reset", 18, 5)):
                x = (d[:, b_] - d[:, a_])[rs]
                print(f"    {n_:34s} mean {x.mean():8.0f} over {rs.sum() / d.shape[0]:.1f} waves per launch")
        print("  wave duration ns: mean %.0f, of the last-finishing wave %.0f" % (((d[:, 14] - d[:, 13]) * 10).mean(), ((d[it, 14, last] - d[it, 13, last]) * 10).mean()))
        print("  last wave to finish (avg over launches), shader cycles per region:")
        for n, a, b in seq:
            print(f"    {n:34s} {(d[it, b, last] - d[it, a, last]).mean():8.0f}")
