"""Development (branch exp/aql-step-path builds only): per-step launches as hand-written AQL packets, with and without an
acquire fence in the packets between the first and the last (RSX_AQL_ACQ=none) and with the per-CU caches invalidated inside the
kernel instead (-DRSX_INKERNEL_INV build).  Raw ctypes: those builds predate the current Python binding.
usage: python tools/exp_aql_acquire.py <lib.so> [label]   (RSX_AQL=1 selects the packet path)"""
import ctypes as C, os, sys, time, hashlib
import numpy as np, torch
lib = C.CDLL(sys.argv[1]); label = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
lib.rsx_last_error.restype = C.c_char_p
vp = C.c_void_p
lib.rsx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
lib.rsx_task_attach.argtypes = [vp, C.c_int, C.c_uint64, C.c_uint64, C.c_int]
lib.rsx_task_reset.argtypes = [vp, vp]; lib.rsx_task_step_n.argtypes = [vp, C.c_int, vp]; lib.rsx_get_state_full.argtypes = [vp, vp, vp]
lib.rsx_destroy.argtypes = [vp]
def chk(rc):
    if rc: raise RuntimeError(lib.rsx_last_error().decode())
s = torch.cuda.current_stream().cuda_stream
for name, kind, ft, nb, ny, task, B, rows in (("vss 4096", 0, 0, 3, 3, 1, 4096, 5 + 6 * 6 + 2), ("sd 2048", 1, 2, 1, 6, 2, 2048, 5 + 11 * 7 + 2), ("11v11 1024", 1, 1, 11, 11, 6, 1024, 5 + 11 * 22 + 2)):
    h = vp(); chk(lib.rsx_create(C.byref(h), kind, ft, nb, ny, 25, B, 0)); chk(lib.rsx_task_attach(h, task, 0, 0, 0)); chk(lib.rsx_task_reset(h, s))
    chk(lib.rsx_task_step_n(h, 2000, s)); torch.cuda.synchronize()
    t = time.perf_counter(); chk(lib.rsx_task_step_n(h, 4000, s)); torch.cuda.synchronize(); us = (time.perf_counter() - t) / 4000 * 1e6
    st = np.empty((B, rows), dtype=np.float64); chk(lib.rsx_get_state_full(h, st.ctypes.data_as(vp), s)); torch.cuda.synchronize()
    print(f"{label:40s} {name:12s} {us:7.2f} us per step   state after 6000 steps: sha1 {hashlib.sha1(st.tobytes()).hexdigest()[:12]}", flush=True)
    lib.rsx_destroy(h)
