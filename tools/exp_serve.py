"""Development: serving (persistent kernel + doorbell) vs per-step launches: equality and step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd.vec import VecVSSEnv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = VecVSSEnv(B, seed=3); b = VecVSSEnv(B, seed=3)
a.reset(); b.reset()
g = torch.Generator(device="cuda"); g.manual_seed(0)
acts = torch.rand(300, B, 2, device="cuda", generator=g) * 2 - 1
torch.cuda.current_stream().synchronize()
b.serve_start(1000)
print("serving", flush=True)
for t in range(300):
    oa = a.step(acts[t]); ob = b.step(acts[t])
    if t % 100 == 99:
        torch.cuda.current_stream().synchronize()
        print(t, "obs equal", torch.equal(oa[0], ob[0]), "rew equal", torch.equal(oa[1], ob[1]), "done equal", torch.equal(oa[2], ob[2]), flush=True)
for env, name in ((a, "launch per step"), (b, "serving")):
    for _ in range(200): env.step(acts[0])
    torch.cuda.current_stream().synchronize(); t0 = time.perf_counter()
    for _ in range(2000): env.step(acts[1])
    torch.cuda.current_stream().synchronize(); print(f"{name}: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us/step", flush=True)
b.serve_stop()
torch.cuda.current_stream().synchronize()
print("state equal after stop", np.array_equal(a.sim.get_state_full(), b.sim.get_state_full()), "metrics", a.metrics()["env_steps"], b.metrics()["env_steps"])
a.close(); b.close()
