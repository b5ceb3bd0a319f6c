"""Python-side cost of VecVSSEnv.step() with device-resident actions (development tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rsoccer_amd.vec import VecVSSEnv
env = VecVSSEnv(4096, seed=0)
obs, _ = env.reset()
act = torch.rand(4096, 2, device="cuda") * 2 - 1
for name, fn in (("step(actions)", lambda: env.step(act)), ("step(None)", lambda: env.step(None)),
                 ("step_random(1)", lambda: env.step_random(1))):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    K = 3000
    t = time.perf_counter()
    for _ in range(K): fn()
    t_issue = time.perf_counter() - t
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t
    print(f"{name:16s} host issue {t_issue / K * 1e6:6.2f} us/step, end-to-end {t_all / K * 1e6:6.2f} us/step")
# a policy-shaped loop: obs -> tiny torch op -> actions
lin = torch.nn.Linear(40, 2).cuda()
with torch.no_grad():
    for _ in range(300): o, r, te, tr, info = env.step(torch.tanh(lin(obs)))
    torch.cuda.synchronize(); K = 2000; t = time.perf_counter()
    for _ in range(K):
        obs, r, te, tr, info = env.step(torch.tanh(lin(obs)))
    torch.cuda.synchronize()
    print(f"policy loop (Linear+tanh -> step): {(time.perf_counter() - t) / K * 1e6:6.2f} us/step")
