"""Regression fixtures of model v2: piles pressed on walls, recorded from THIS project's float64 oracle (not a reference
pin — the reference has no physics of its own; tests/test_physics_fence.py pins the arrays by hash and checks that the oracle still
reproduces them, so a model edit is visible).  Needs nothing but the oracle:  python tests/golden/make_model_v2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O


def scrum(seed, steps=300, every=20):
    """22 SSL robots chasing the ball on the division-A field (the scenario of test_crowded_11v11_full_size_contact_invariants), the
    ball started near a corner so that the pile is at the walls within a few seconds"""
    N = 22
    rng = np.random.default_rng(seed)
    s = O.OracleEnv(1, 1, 11, 11, 25, "f64")
    grid = np.array([(0.2 * (i - 2.5), 0.2 * (j - 1.5)) for i in range(6) for j in range(4)][:N]) + (5.0, 3.6)
    pose = np.zeros((N, 3)); pose[:, :2] = grid + rng.uniform(-0.008, 0.008, (N, 2)); pose[:, 2] = rng.uniform(-180, 180, N)
    s.reset(np.array([5.6, 4.3, 0.5, 0.5]), pose[:11], pose[11:])
    reset_state = s.get_state_full().copy()
    cmds, states = [], []
    for t in range(steps):
        st = s.get_state()
        x, y, th = st[5::11][:N], st[6::11][:N], np.deg2rad(st[7::11][:N])
        gx, gy = st[0] - x, st[1] - y
        n = np.hypot(gx, gy) + 1e-9
        gx, gy = 2.0 * gx / n, 2.0 * gy / n
        cm = np.zeros((N, 8))
        cm[:, 1] = gx * np.cos(th) + gy * np.sin(th); cm[:, 2] = -gx * np.sin(th) + gy * np.cos(th)
        cm[:, 3] = rng.uniform(-3, 3, N); cm[:, 5] = (rng.uniform(size=N) > 0.9) * 3.0; cm[:, 7] = rng.uniform(size=N) > 0.5
        cm = cm.astype(np.float32).astype(np.float64)     # stored as float32: the run itself takes the rounded values
        s.step(cm)
        cmds.append(cm)
        if t % every == every - 1:
            states.append(s.get_state_full().copy())
    return reset_state, np.array(cmds), np.array(states)


def main():
    O.build()
    out = {}
    for k, seed in enumerate((11, 12)):
        r, c, st = scrum(seed)
        out[f"scrum{k}_reset_state"], out[f"scrum{k}_cmds"], out[f"scrum{k}_states"] = r, c.astype(np.float32), st
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_v2_piles.npz"), **out)
    for k in sorted(out):
        print(k, out[k].shape, out[k].dtype)


if __name__ == "__main__":
    main()
