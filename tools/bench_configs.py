"""Measures every BASELINE.json config that fits one GPU (parity-tested configs; bench.py is the
contract for configs[1]) plus a batch-size sweep.  Writes a markdown table.
usage: python tools/bench_configs.py [out.md]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rsoccer_amd import _lib as L

s = torch.cuda.current_stream().cuda_stream
rows = []


def timeit(fn, K):
    fn(max(20, K // 10)); torch.cuda.synchronize()
    t = time.perf_counter(); fn(K); torch.cuda.synchronize()
    return (time.perf_counter() - t) / K * 1e6


def fused(name, kind, ft, nb, ny, task, B, bytes_per_step, K=2000):
    sim = L.Sim(kind, ft, nb, ny, 25, B); sim.task_attach(task, 0, 0, 0); sim.task_reset()
    us = timeit(lambda n: sim.task_step_n(n, s), K)
    ur = timeit(lambda n: sim.task_rollout(n, s), K)
    rows.append((name, B, us, B / us, bytes_per_step * B / us / 1e3, ur, B / ur))
    sim.close()


def raw_ssl(name, ft, nb, ny, B, crowded, bytes_per_step, K=1000):
    """SSL raw simulator (rsx_step_dev) with per-step random local-velocity commands written by
    torch into the SoA command buffer — BASELINE configs[3] (synthetic SSLBaseEnv-style task)."""
    sim = L.Sim(1, ft, nb, ny, 25, B)
    N = nb + ny
    rng = np.random.default_rng(0)
    f = sim.get_field_params()
    ball = np.zeros((B, 4)); rob = np.zeros((B, N, 3))
    if crowded:   # every robot inside a 1.5 m disc around the ball: saturates the contact sweep
        for k in range(N):
            r, a = 0.25 + 1.25 * np.sqrt((k + 0.5) / N), 2.4 * k
            rob[:, k, 0] = r * np.cos(a); rob[:, k, 1] = r * np.sin(a)
    else:         # jittered grid, >= 0.2 m apart
        gx, gy = np.meshgrid(np.linspace(-4.5, 4.5, 6), np.linspace(-3.2, 3.2, 4))
        pts = np.stack([gx.ravel(), gy.ravel()], 1)[:N]
        rob[:, :, :2] = pts[None] + rng.uniform(-0.3, 0.3, (B, N, 2))
        ball[:, :2] = rng.uniform(-0.2, 0.2, (B, 2)) + [0.9, 0.8]
    rob[:, :, 2] = rng.uniform(-180, 180, (B, N))
    sim.reset(ball, rob[:, :nb], rob[:, nb:])
    cm = sim.cmds_tensor().view(N, 8, B)
    scale = torch.tensor([2.5, 2.5, 10.0], device="cuda").view(1, 3, 1)

    def run(n):
        for _ in range(n):
            cm[:, 1:4, :] = (torch.rand(N, 3, B, device="cuda") * 2 - 1) * scale
            sim.step_dev(s)
    us = timeit(run, K)
    def run_sim_only(n):
        for _ in range(n):
            sim.step_dev(s)
    uso = timeit(run_sim_only, K)
    rows.append((name + " (torch command generation included)", B, us, B / us, bytes_per_step * B / us / 1e3, float("nan"), float("nan")))
    rows.append((name + " (step kernel only)", B, uso, B / uso, bytes_per_step * B / uso / 1e3, float("nan"), float("nan")))
    sim.close()


def raw_random(name, kind, ft, nb, ny, B, bytes_per_step, K=1000):
    """raw simulator with commands drawn in the kernel (rsx_step_dev_random): no torch op in the loop"""
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    sim.task_attach(L.TASK_SSL_SCRIMMAGE_CROWDED if "crowded" in name else L.TASK_SSL_SCRIMMAGE, 0, 0, 0)
    sim.task_reset()                      # the scrimmage line-up, then the raw kernel takes over
    tick = [0]
    def run(n):
        sim.step_dev_random(n, 1, tick[0], s); tick[0] += n
    us = timeit(run, K)
    rows.append((name, B, us, B / us, bytes_per_step * B / us / 1e3, float("nan"), float("nan")))
    sim.close()


SCRIM_BYTES = 2 * 4 * (5 + 11 * 22) + 4 * 8 * 22 + 4 * (2 + 2 * 22) + 4 + 1   # state r/w + commands + obs + reward + done = 2869 (bench.py's figure; the task draws its commands on the device, so 704 of these bytes are not moved)
fused("configs[1] VSS-v0 3v3 fused", 0, 0, 3, 3, 1, 4096, 541)
fused("configs[2] SSLStaticDefenders-v0 1v6 fused", 1, 2, 1, 6, 2, 2048, 981)
fused("SSLDribbling-v0 1v4 fused", 1, 2, 1, 4, 3, 2048, 2 * 4 * 60 + 4 * 40 + 4 * 21 + 5)
fused("SSLContestedPossession-v0 1v1 fused", 1, 2, 1, 1, 4, 2048, 2 * 4 * 27 + 4 * 16 + 4 * 14 + 5)
fused("SSLPassEndurance-v0 2v0 fused", 1, 2, 2, 0, 5, 2048, 2 * 4 * 27 + 4 * 16 + 4 * 16 + 5)
fused("VSS-v0 on the 5v5 field (field_type 1) fused", 0, 1, 5, 5, 1, 4096, 2 * 4 * 66 + 4 * 20 + 4 * 64 + 5)
fused("configs[3] SSL 11v11 scrimmage task (fused, every robot commanded), spread", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE, 1024, SCRIM_BYTES)
fused("configs[3] SSL 11v11 scrimmage task (fused), crowded: worst-case contacts", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE_CROWDED, 1024, SCRIM_BYTES)
fused("configs[3] SSL 11v11 scrimmage task (fused), spread", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE, 65536, SCRIM_BYTES, K=200)
fused("configs[3] SSL 11v11 scrimmage task (fused), crowded", 1, 1, 11, 11, L.TASK_SSL_SCRIMMAGE_CROWDED, 65536, SCRIM_BYTES, K=200)
raw_random("configs[3] SSL 11v11 raw sim, device-drawn commands, spread", 1, 1, 11, 11, 1024, 2680)
raw_random("configs[3] SSL 11v11 raw sim, device-drawn commands, crowded", 1, 1, 11, 11, 1024, 2680)
raw_random("configs[3] SSL 11v11 raw sim, device-drawn commands, spread", 1, 1, 11, 11, 65536, 2680, K=200)
raw_random("configs[3] SSL 11v11 raw sim, device-drawn commands, crowded", 1, 1, 11, 11, 65536, 2680, K=200)
raw_ssl("configs[3] SSL 11v11 raw sim, spread", 1, 11, 11, 1024, False, 2680)
raw_ssl("configs[3] SSL 11v11 raw sim, crowded (worst-case contacts)", 1, 11, 11, 1024, True, 2680)
for B in (256, 1024, 16384, 65536, 262144, 1048576, 4194304):
    fused("sweep VSS-v0 fused", 0, 0, 3, 3, 1, B, 541, K=2000 if B <= 65536 else 200)
for B in (16384, 262144, 1048576):
    fused("sweep SSLStaticDefenders-v0 fused", 1, 2, 1, 6, 2, B, 981, K=500 if B <= 65536 else 100 if B <= 262144 else 60)
for B in (262144, 1048576):   # the other registered SSL tasks at scale (one lane per env from 65 536 envs)
    fused("sweep SSLDribbling-v0 fused", 1, 2, 1, 4, 3, B, 2 * 4 * 60 + 4 * 40 + 4 * 21 + 5, K=60)
    fused("sweep SSLContestedPossession-v0 fused", 1, 2, 1, 1, 4, B, 2 * 4 * 27 + 4 * 16 + 4 * 14 + 5, K=60)
    fused("sweep SSLPassEndurance-v0 fused", 1, 2, 2, 0, 5, B, 2 * 4 * 27 + 4 * 16 + 4 * 16 + 5, K=60)

# The robosim-shaped host path (rsx_step + rsx_get_state): float64 host arrays in and out, i.e.
# the PCIe-inclusive rate of the boundary when a caller keeps its data on the host.
host_rows = []
for kind, ft, nb, ny, B in ((0, 0, 3, 3, 1), (0, 0, 3, 3, 4096), (1, 2, 1, 6, 2048)):
    sim = L.Sim(kind, ft, nb, ny, 25, B)
    N, C = nb + ny, (2 if kind == 0 else 8)
    cmds = np.random.default_rng(1).uniform(-1, 1, (B, N, C)) * (30.0 if kind == 0 else 1.0)
    if kind == 1:
        cmds[..., 0] = 0.0; cmds[..., 5:] = 0.0
    def run(n):
        for _ in range(n):
            sim.step(cmds); sim.get_state()
    us = timeit(run, 300)
    host_rows.append((("VSS 3v3" if kind == 0 else "SSL 1v6"), B, us, B / us))
    sim.close()

lines = ["| config | envs | us / step (1 launch per step) | M env-steps/s | algorithmic GB/s | frac of 8 TB/s | us / step (one launch) | M env-steps/s (one launch) |",
         "|---|---|---|---|---|---|---|---|"]
for name, B, us, rate, gbs, ur, rr in rows:
    lines.append(f"| {name} | {B} | {us:.2f} | {rate:.1f} | {gbs:.1f} | {gbs / 8000 * 100:.2f} % | {ur:.2f} | {rr:.1f} |")
lines += ["", "Host-format path (`rsx_step` + `rsx_get_state`, float64 host arrays, synchronous, PCIe both ways):", "",
          "| simulator | envs | us / step+get_state | M env-steps/s |", "|---|---|---|---|"]
for name, B, us, rate in host_rows:
    lines.append(f"| {name} | {B} | {us:.1f} | {rate:.3f} |")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("# Per-config and batch-sweep measurements (MI355X)\n\nProduced by `python tools/bench_configs.py` "
                                 "(algorithmic bytes per env-step: SURVEY.md 8(d)).\n\n" + text + "\n")
