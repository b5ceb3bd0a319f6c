"""Regression fixtures of model v2 (SSL: round 5, VSS: round 6): piles pressed on walls, recorded from THIS project's float64 oracle (not a reference
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


def vss_scrum(seed, steps=400, every=20):
    """six VSS robots driving at the ball on the 1.5 m x 1.3 m field (the scenario of test_vss_scrum_full_size_contact_invariants), the
    ball started near a goal mouth: within a second the pile is at the goal line's wall, at a post and in the goal box (round 6:
    held axes + chord posts)"""
    N = 6
    rng = np.random.default_rng(seed)
    s = O.OracleEnv(0, 0, 3, 3, 25, "f64")
    pose = np.zeros((N, 3))
    pose[:, 0] = 0.25 + 0.11 * (np.arange(N) % 3) + rng.uniform(-0.01, 0.01, N)
    pose[:, 1] = np.where(np.arange(N) < 3, 0.12, -0.12) + rng.uniform(-0.01, 0.01, N)
    pose[:, 2] = rng.uniform(-180, 180, N)
    s.reset(np.array([0.66, 0.17 if seed % 2 else -0.05, 0.3, 0.0]), pose[:3], pose[3:])
    reset_state = s.get_state_full().copy()
    cmds, states = [], []
    for t in range(steps):
        st = s.get_state()
        x, y, th = st[5::6][:N], st[6::6][:N], np.deg2rad(st[7::6][:N])
        err = np.arctan2(st[1] - y, st[0] - x) - th
        err = np.arctan2(np.sin(err), np.cos(err))
        v = 0.9 * np.maximum(np.cos(err), 0.0) + 0.1
        w = 8.0 * err + rng.uniform(-2, 2, N)
        cm = np.stack([(v - w * 0.04) / 0.026, (v + w * 0.04) / 0.026], 1)
        cm = cm.astype(np.float32).astype(np.float64)     # stored as float32: the run itself takes the rounded values
        s.step(cm)
        cmds.append(cm)
        if t % every == every - 1:
            states.append(s.get_state_full().copy())
    return reset_state, np.array(cmds), np.array(states)


def main():
    O.build()
    here = os.path.dirname(os.path.abspath(__file__))
    out = {}
    for k, seed in enumerate((11, 12)):
        r, c, st = scrum(seed)
        out[f"scrum{k}_reset_state"], out[f"scrum{k}_cmds"], out[f"scrum{k}_states"] = r, c.astype(np.float32), st
    np.savez_compressed(os.path.join(here, "model_v2_piles.npz"), **out)
    for k in sorted(out):
        print(k, out[k].shape, out[k].dtype)
    out = {}
    for k, seed in enumerate((21, 22)):
        r, c, st = vss_scrum(seed)
        out[f"vss{k}_reset_state"], out[f"vss{k}_cmds"], out[f"vss{k}_states"] = r, c.astype(np.float32), st
    np.savez_compressed(os.path.join(here, "model_v2_vss_piles.npz"), **out)
    for k in sorted(out):
        print(k, out[k].shape, out[k].dtype)


if __name__ == "__main__":
    main()
