"""Development: the PCIe-inclusive rate of the robosim-shaped boundary — rsx_step_state with host float64 arrays in the reference's wire
format (commands in, state out: host AoS f64 <-> device SoA f32 conversion, both copies, one synchronisation per step), VSS 3v3 raw
simulator, by batch size; and the fused VSS-v0 task stepped with host-resident actions and observations read back every step.
python tools/exp_host_boundary.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rsoccer_amd import _lib as L
rng = np.random.default_rng(0)
for B in (1, 64, 4096, 65536):
    sim = L.Sim(0, 0, 3, 3, 25, B)
    cmds = np.ascontiguousarray(rng.uniform(-30, 30, (B, 6, 2)))
    n = 2000 if B <= 4096 else 200
    for _ in range(20): sim.step_state(cmds)
    t = time.perf_counter()
    for _ in range(n): st = sim.step_state(cmds)
    dt = (time.perf_counter() - t) / n
    print(f"raw VSS 3v3, host wire format (rsx_step_state), {B:6d} envs: {dt * 1e6:9.1f} us per step = {B / dt:12.4g} env-steps/s", flush=True)
    wire = sim.wire_buffers()
    if wire is not None:   # ABI 6: the caller works in the handle's pinned wire buffers — no pass of a CPU thread over the data at all
        wire[0][...] = cmds
        for _ in range(20): sim.step_wire()
        t = time.perf_counter()
        for _ in range(n): sim.step_wire()
        dt = (time.perf_counter() - t) / n
        print(f"raw VSS 3v3, pinned wire buffers (rsx_step_wire),  {B:6d} envs: {dt * 1e6:9.1f} us per step = {B / dt:12.4g} env-steps/s", flush=True)
    sim.close()
B = 4096
sim = L.Sim(0, 0, 3, 3, 25, B); sim.task_attach(1, 0, 0, 0); sim.task_reset(); tens = sim.task_tensors()
a_host = torch.from_numpy(rng.uniform(-1, 1, (B, 2)).astype(np.float32)).pin_memory()
obs_host = torch.empty((B, 40), dtype=torch.float32).pin_memory(); rew_host = torch.empty(B, dtype=torch.float32).pin_memory(); fl_host = torch.empty((2, B), dtype=torch.uint8).pin_memory()
def step():
    tens["actions"].copy_(a_host, non_blocking=True)
    sim.task_step(tens["actions"].data_ptr())
    obs_host.copy_(tens["obs"], non_blocking=True); rew_host.copy_(tens["reward"], non_blocking=True)
    fl_host[0].copy_(tens["terminated"], non_blocking=True); fl_host[1].copy_(tens["truncated"], non_blocking=True)
    torch.cuda.synchronize()
for _ in range(50): step()
t = time.perf_counter()
for _ in range(2000): step()
dt = (time.perf_counter() - t) / 2000
print(f"fused VSS-v0, {B} envs, actions from / observations + reward + flags to pinned host memory every step: {dt * 1e6:7.1f} us per step = {B / dt:10.4g} env-steps/s", flush=True)
