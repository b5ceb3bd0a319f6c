"""The jammed pile of BASELINE.json configs[3] (22 SSL robots all driving at the ball) on the CPU oracle: how deep do
robots overlap, and what would more contact sweeps or a position-only projection pass buy?  (Model EXPERIMENTS:
RSXO_SWEEPS / RSXO_PROJECT are off in the model of DESIGN.md 4.)   python tools/exp_jam.py [envs] [steps]"""
import os, subprocess, sys, time
CHILD = r'''
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle as O
B, T, N = int(sys.argv[1]), int(sys.argv[2]), 22
O.build(); O.set_threads(min(16, os.cpu_count() or 1))
rng = np.random.default_rng(3)
grid = np.array([(0.2 * (i - 2.5), 0.2 * (j - 1.5)) for i in range(6) for j in range(4)][:N])
envs = []
for e in range(B):
    s = O.OracleEnv(1, 1, 11, 11, 25, "f32")
    pose = np.zeros((N, 3)); pose[:, :2] = grid + rng.uniform(-0.008, 0.008, (N, 2)); pose[:, 2] = rng.uniform(-180, 180, N)
    s.reset(np.array([0.0, 0.1, 0.0, 0.0]), pose[:11], pose[11:])
    envs.append(s)
worst, hist = 0.0, []
t0 = time.perf_counter()
for t in range(T):
    for s in envs:
        st = s.get_state()
        x, y, th = st[5::11][:N], st[6::11][:N], np.deg2rad(st[7::11][:N])
        gx, gy = st[0] - x, st[1] - y
        n = np.hypot(gx, gy) + 1e-9
        gx, gy = 2.0 * gx / n, 2.0 * gy / n
        cm = np.zeros((N, 8))
        cm[:, 1] = gx * np.cos(th) + gy * np.sin(th); cm[:, 2] = -gx * np.sin(th) + gy * np.cos(th)
        cm[:, 3] = rng.uniform(-3, 3, N); cm[:, 5] = (rng.uniform(size=N) > 0.9) * 3.0; cm[:, 7] = rng.uniform(size=N) > 0.5
        s.step(cm)
        if t % 5 == 4 and t > 100:
            st = s.get_state(); x, y = st[5::11][:N], st[6::11][:N]
            d = np.hypot(x[:, None] - x[None], y[:, None] - y[None]) + 9.0 * np.eye(N)
            ov = 0.18 - d.min(1)
            hist.append(ov.max()); worst = max(worst, ov.max())
dt = time.perf_counter() - t0
h = np.array(hist)
print(f"sweeps<={os.environ.get('RSXO_SWEEPS','2'):>2s} project={os.environ.get('RSXO_PROJECT','0'):>2s}: worst overlap {100*worst:5.2f} cm, "
      f"median of the per-env worst {100*np.median(h):5.2f} cm, p99 {100*np.percentile(h,99):5.2f} cm; {1e6*dt/(B*T):7.1f} us per env-step (one CPU thread, incl. the Python loop)", flush=True)
'''
args = sys.argv[1:3] if len(sys.argv) >= 3 else ["24", "600"]
for sw, pr in (("2", "0"), ("4", "0"), ("8", "0"), ("16", "0"), ("2", "2"), ("2", "8"), ("2", "32")):
    subprocess.run([sys.executable, "-c", CHILD] + args, env=dict(os.environ, RSXO_SWEEPS=sw, RSXO_PROJECT=pr))
