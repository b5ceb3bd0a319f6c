"""Development: where the serving round trip spends its time (doorbell write vs completion wait)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd.vec import VecVSSEnv
B = 4096
b = VecVSSEnv(B, seed=3); b.reset()
act = torch.rand(B, 2, device="cuda") * 2 - 1
st = torch.cuda.current_stream()
st.synchronize()
b.serve_start(3000)
for _ in range(50): b.step(act)
st.synchronize(); t0 = time.perf_counter()
for _ in range(1000): b.step(act)
st.synchronize(); dt = time.perf_counter() - t0
print(f"serve step ({'no wait' if os.environ.get('RSX_SERVE_NOWAIT') else 'write + wait'}): {dt / 1000 * 1e6:.2f} us/step", flush=True)
time.sleep(0.2)
b.serve_stop()
print("env_steps", b.metrics()["env_steps"])
b.close()
