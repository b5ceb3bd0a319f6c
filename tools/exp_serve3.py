"""Development: how long does an un-rung serving kernel live (watchdog check)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd.vec import VecVSSEnv
b = VecVSSEnv(256, seed=3); b.reset(); torch.cuda.synchronize()
for ms in (100, 300, 1000):
    t0 = time.perf_counter()
    b.sim.serve_start(ms)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"timeout {ms} ms: start call {1e3 * (t1 - t0):.1f} ms, kernel lived {1e3 * (time.perf_counter() - t1):.1f} ms", flush=True)
    b.sim.serve_stop()
b.close()
