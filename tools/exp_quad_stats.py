"""Development: what the contact path of the four-lanes-per-env 11v11 kernel does per sub-step (counters of a
-DRSX_QSTATS build: tools/build_variant.sh qstats -DRSX_QSTATS; RSX_LIB=tools/_dev/librsx_qstats.so RSX_LAYOUT=quad)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from rsoccer_amd import _lib as L
lib = L.load()
lib.rsx_debug_qstats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
s = torch.cuda.current_stream().cuda_stream
NAMES = ["wave sub-steps", "sweeps on the contact path", "screening passes", "partner sets (slots)", "slots walked", "walk trips (sum over slots of max over lanes)",
         "contacts (robot sides)", "ball passes", "ball walk trips", "walk trips if flattened (max over lanes of the lane's total)", "hot sweeps", "second sweeps", "robots near ball", "steps x5: closest pair of the wave > 26 cm", "steps x5: > 30 cm", "steps x5: > 34 cm"]
B = int(os.environ.get("B", "65536"))
for task, name in ((6, "spread"), (7, "crowded")):
    sim = L.Sim(1, 1, 11, 11, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
    for lo, n in ((0, 20), (20, 60), (80, 200), (280, 400)):
        out = (C.c_ulonglong * 16)()
        lib.rsx_debug_qstats(out, 1)
        torch.cuda.synchronize(); t = time.perf_counter()
        sim.task_step_n(n, s)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n * 1e6
        lib.rsx_debug_qstats(out, 1)
        w = max(1, out[0])
        print(f"== 11v11 {name} {B} envs, steps {lo}..{lo + n}: {dt:.1f} us per step (counters on), layout {sim.task_layout()}")
        for i, nm in enumerate(NAMES):
            print(f"   {nm:62s} {out[i] / w:8.3f} per wave sub-step")
    sim.close()
