"""The scrum of tests/test_gpu_parity.py::test_crowded_11v11_full_size_contact_invariants on the CPU oracle (22 SSL robots chasing
the ball on the division-A field, random kicks / dribblers, 1000 steps): how deep do robots overlap, and where do the deep
overlaps sit (at a wall / goal or in the open)?  Round 5 used it with experiment knobs to pick model v2 (profiles/r05_jam_model_v2.txt).
    python tools/exp_jam2.py [envs] [steps]"""
import os, subprocess, sys
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
f = envs[0].field_params()
XL, YL = f[0] / 2 + 0.3 - 0.09, f[1] / 2 + 0.3 - 0.09
GB = f[0] / 2 + f[5]            # goal back wall
wall_s, open_s, all_s = [], [], []
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
        if t % 5 == 4:
            st = s.get_state(); x, y = st[5::11][:N], st[6::11][:N]
            d = np.hypot(x[:, None] - x[None], y[:, None] - y[None]) + 9.0 * np.eye(N)
            i, j = np.unravel_index(np.argmin(d), d.shape)
            ov = max(0.0, 0.18 - d[i, j])
            near = lambda k: abs(x[k]) > XL - 0.25 or abs(y[k]) > YL - 0.25 or (abs(x[k]) > f[0] / 2 - 0.1 and abs(y[k]) < f[4] / 2 + 0.3)
            (wall_s if (near(i) or near(j)) else open_s).append(ov)
            all_s.append(ov)
dt = time.perf_counter() - t0
a, w, o = np.array(all_s), np.array(wall_s + [0.0]), np.array(open_s + [0.0])
print(f"worst {100*a.max():5.2f} cm, p99 {100*np.percentile(a,99):5.2f} cm | "
      f"samples near a wall / goal: {len(wall_s):6d} worst {100*w.max():5.2f} p99 {100*np.percentile(w,99):5.2f} | in the open: {len(open_s):6d} worst {100*o.max():5.2f} p99 {100*np.percentile(o,99):5.2f}", flush=True)
'''
args = sys.argv[1:3] if len(sys.argv) >= 3 else ["32", "1000"]
subprocess.run([sys.executable, "-c", CHILD] + args)
