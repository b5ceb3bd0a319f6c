"""The fence around the (frozen) 2-D step model, DESIGN.md section 4 — checks that do not depend on how the model is
implemented:

  * an isolated impact conserves linear momentum to round-off (robot - ball and robot - robot, head-on and oblique,
    with the friction impulse), in the three backends;
  * refining the sub-step changes a contact-free and a single-contact trajectory by O(h);
  * the physics-regression arrays under tests/golden/ are pinned by hash: together with
    test_oracle_golden.py::test_physics_regression_against_recorded_oracle_runs (the oracle reproduces them) this
    makes a model edit visible — changing the model breaks that test, regenerating the arrays breaks this one.
"""
import hashlib
import math
import os

import numpy as np
import pytest

from test_physics_model import BACKENDS, _cmd, _make

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz")
# sha256 over name, dtype, shape and bytes of every '*_states' / '*_reset_state' array (sorted by name)
REGRESSION_SHA256 = "3828b6fb045bf749a0d757730e22de5068203e14e8283536e916b3a1052e032f"

M_ROBOT = {0: 0.18, 1: 2.2}
M_BALL = 0.046
MU_G = {0: 0.3, 1: 0.4}


def test_physics_regression_arrays_are_frozen():
    z = np.load(GOLDEN)
    h = hashlib.sha256()
    n = 0
    for k in sorted(z.files):
        if "_states" in k or "reset_state" in k:
            a = np.ascontiguousarray(z[k])
            h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
            n += 1
    assert n == 59
    assert h.hexdigest() == REGRESSION_SHA256, "the recorded physics trajectories changed: the model of DESIGN.md 4 is frozen"


def _tol(backend, t64, t32):
    return t64 if backend == "f64" else t32


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("kind,offset", [(0, 0.0), (0, 0.02), (1, 0.0), (1, 0.05)], ids=["vss-head-on", "vss-oblique", "ssl-head-on", "ssl-oblique"])
def test_robot_ball_impact_conserves_linear_momentum(backend, oracle_mod, kind, offset):
    """A ball hits the BACK of a resting robot (body circle; `offset` makes the hit oblique, so the Coulomb impulse
    acts too).  5 ms steps = one sub-step per step(): the state after the step of the impact is the state right after
    the impulses.  What the robot gained is what the ball lost, as a vector: m_r dv_r + m_b (v_b' - v_b_free) = 0,
    v_b_free = the ball's velocity had it only felt the rolling resistance of that step."""
    ts = 5
    r_sum = (0.0375 if kind == 0 else 0.09) + 0.0215
    if kind == 0:
        s = _make(backend, 0, 0, 3, 3, ts)
        blue = [[0.0, 0.0, 0.0], [-0.6, 0.5, 0.0], [-0.6, -0.5, 0.0]]
        yel = [[0.6, 0.5, 0.0], [0.6, 0.3, 0.0], [0.6, -0.5, 0.0]]
        n_cmd = (6, 2)
    else:
        s = _make(backend, 1, 2, 1, 0, ts)
        blue, yel, n_cmd = [[0.0, 0.0, 0.0]], np.zeros((0, 3)), (1, 8)
    v0 = 1.5
    s.reset(np.array([-(r_sum + 0.05), offset, v0, 0.0]), np.array(blue, float), np.array(yel, float))
    hit = False
    for _ in range(40):
        before = s.get_state()
        s.step(np.zeros(n_cmd))
        after = s.get_state()
        dv_r = after[8:10] - before[8:10]
        if np.any(dv_r != 0.0):
            hit = True
            vb = before[3:5]
            sp = np.hypot(*vb)
            free = vb * max(0.0, sp - MU_G[kind] * ts * 1e-3) / sp          # rolling resistance of this step
            dp = M_ROBOT[kind] * dv_r + M_BALL * (after[3:5] - free)
            scale = M_BALL * v0
            assert np.all(np.abs(dp) < _tol(backend, 1e-12, 3e-6) * scale), (dp / scale, dv_r, after[3:5], free)
            assert dv_r[0] > 0.0 and after[3] < free[0]                    # the robot was pushed, the ball slowed down
            if offset:
                assert dv_r[1] != 0.0                                      # oblique: a sideways component was exchanged
            break
    assert hit, "the ball never reached the robot"


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("offset", [0.0, 0.06], ids=["head-on", "oblique"])
def test_robot_robot_impact_conserves_linear_momentum(backend, oracle_mod, offset):
    """Two holonomic robots of equal mass; A is commanded to hold exactly the velocity it has (its actuation changes
    nothing), B rests with a zero command.  One sub-step per step(): after the step of the impact
    m (dv_A + dv_B) = 0 to round-off."""
    ts = 5
    s = _make(backend, 1, 1, 1, 1, ts)
    v0 = 1.0
    s.reset(np.array([3.0, 3.0, 0.0, 0.0]), np.array([[-0.30, offset, 0.0]]), np.array([[0.0, 0.0, 0.0]]))
    cmd = _cmd(2, 8, {0: [0, v0, 0.0, 0.0]})
    for _ in range(60):                                     # ramp up to v0 well before the contact
        s.step(cmd)
        st = s.get_state()
        if abs(st[5 + 3] - v0) < 1e-6:
            break
    hit = False
    for _ in range(80):
        before = s.get_state()
        s.step(cmd)
        after = s.get_state()
        va0, vb0 = before[8:10], before[19:21]
        va1, vb1 = after[8:10], after[19:21]
        if np.any(vb1 != vb0):
            hit = True
            dp = (va1 - va0) + (vb1 - vb0)
            assert np.all(np.abs(dp) < _tol(backend, 1e-12, 3e-6) * v0), (dp, va0, va1, vb1)
            assert vb1[0] > 0.0 and va1[0] < va0[0]
            break
    assert hit, "the robots never met"


def _trajectory(backend, kind, ts, horizon_ms, setup, command):
    s = setup(ts)
    for _ in range(horizon_ms // ts):
        s.step(command)
    return s.get_state()


@pytest.mark.parametrize("backend", BACKENDS)
def test_refining_the_sub_step_changes_a_contact_free_trajectory_by_order_h(backend, oracle_mod):
    """A differential-drive robot accelerating along an arc and a rolling ball, no contact, 160 ms.  time_step_ms 4 / 2 /
    1 are one sub-step of h = 4 / 2 / 1 ms per step() (the per-step terms — rolling resistance, command latch — scale
    with the step).  First-order integration: the change from h to h / 2 halves as h halves."""
    def setup(ts):
        s = _make(backend, 0, 0, 3, 3, ts)
        s.reset(np.array([0.3, 0.4, -0.6, 0.3]), np.array([[-0.4, -0.2, 20.0], [-0.6, 0.5, 0.0], [-0.6, -0.5, 0.0]]),
                np.array([[0.6, 0.5, 0.0], [0.6, 0.0, 0.0], [0.6, -0.5, 0.0]]))
        return s
    cmd = _cmd(6, 2, {0: [15.0, 25.0]})
    a, b, c = (_trajectory(backend, 0, ts, 160, setup, cmd) for ts in (4, 2, 1))
    for idx, name in ((slice(5, 7), "robot position"), (slice(0, 2), "ball position")):
        d1, d2 = np.linalg.norm(a[idx] - b[idx]), np.linalg.norm(b[idx] - c[idx])
        assert d1 < 4e-3, (name, d1)                         # millimetres at h = 4 ms
        if name == "robot position":
            assert 1.4 < d1 / d2 < 2.8, (name, d1, d2)       # ~ K h: halves with h
        else:
            assert d1 < 2e-4 and d2 < 2e-4, (name, d1, d2)   # the ball's constant deceleration is integrated almost exactly
    th = [x[7] for x in (a, b, c)]
    assert abs(th[0] - th[1]) < 1.0 and abs(th[1] - th[2]) < abs(th[0] - th[1]) + 1e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_refining_the_sub_step_changes_a_single_contact_trajectory_by_order_h(backend, oracle_mod):
    """A ball bounces once off the back of a resting SSL robot.  The impact is detected up to v_rel * h late, so the
    positions afterwards differ by O(h) — bounded by a few v * h, and shrinking as h does; the rebound speed (set by
    the impulse law, not by h) agrees closely."""
    def setup(ts):
        s = _make(backend, 1, 2, 1, 0, ts)
        s.reset(np.array([-0.45, 0.0, 2.0, 0.0]), np.array([[0.0, 0.0, 0.0]]), np.zeros((0, 3)))
        return s
    cmd = np.zeros((1, 8))
    a, b, c = (_trajectory(backend, 1, ts, 320, setup, cmd) for ts in (4, 2, 1))
    for x in (a, b, c):
        assert x[3] < 0.0                                    # it bounced in all three
    v = 2.0
    d1, d2 = abs(a[0] - b[0]), abs(b[0] - c[0])
    assert d1 < 3 * v * 4e-3 and d2 < 3 * v * 2e-3, (d1, d2)
    assert abs(a[3] - c[3]) < 0.05 * abs(c[3]), (a[3], c[3])
    assert abs(a[5] - c[5]) < 2e-3                           # the robot was pushed (then braked by its motors) the same way


# ---- model v2 (round 5): piles pressed on walls, recorded from the float64 oracle by tests/golden/make_model_v2.py ----
V2_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_v2_piles.npz")
V2_SHA256 = "c2330bf6be7dc3dc9e75242f533f79a5a94a71b29d5d25b6c91fbfa9a09a0943"


def test_model_v2_regression_arrays_are_frozen():
    z = np.load(V2_GOLDEN)
    h = hashlib.sha256()
    for k in sorted(z.files):
        a = np.ascontiguousarray(z[k])
        h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    assert len(z.files) == 6
    assert h.hexdigest() == V2_SHA256, "the recorded wall-pile trajectories changed: model v2 of DESIGN.md 4 is frozen"


@pytest.mark.parametrize("k", [0, 1])
def test_model_v2_piles_are_reproduced_and_are_piles_at_walls(oracle_mod, k):
    """the oracle reproduces the recorded 22-robot scrums (300 steps each, sampled every 20th), and the recordings ARE what they are
    for: a pile at the walls (some robot at a boundary limit) that stays a pile of discs (overlap below 1 cm) — the wall-aware shares,
    the third / fourth sweep and the wall clamp all take part in every sample"""
    z = np.load(V2_GOLDEN)
    e = oracle_mod.OracleEnv(1, 1, 11, 11, 25, "f64")
    e.set_state_full(z[f"scrum{k}_reset_state"])
    cmds, want = z[f"scrum{k}_cmds"].astype(np.float64), z[f"scrum{k}_states"]
    f = e.field_params()
    xl, yl = f[0] / 2 + 0.3 - 0.09, f[1] / 2 + 0.3 - 0.09
    at_wall = 0
    for t in range(len(cmds)):
        e.step(cmds[t])
        if t % 20 == 19:
            got = e.get_state_full()
            assert np.allclose(got, want[t // 20], rtol=0, atol=1e-12), (k, t)
            x, y = got[5::11][:22], got[6::11][:22]
            d = np.hypot(x[:, None] - x[None], y[:, None] - y[None]) + 9.0 * np.eye(22)
            assert 0.18 - d.min() < 0.01
            at_wall += int((np.abs(x).max() > xl - 1e-3) or (np.abs(y).max() > yl - 1e-3))
    assert at_wall >= 10


# ---- model v2 of the VSS class (round 6): scrums at a goal mouth, recorded from the float64 oracle by tests/golden/make_model_v2.py ----
V2_VSS_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "model_v2_vss_piles.npz")
V2_VSS_SHA256 = "8796b6f89faa70132f1feb8874b233e4f321b6e103f45bb3a223086ca98f55d4"


def test_model_v2_vss_regression_arrays_are_frozen():
    z = np.load(V2_VSS_GOLDEN)
    h = hashlib.sha256()
    for k in sorted(z.files):
        a = np.ascontiguousarray(z[k])
        h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode()); h.update(a.tobytes())
    assert len(z.files) == 6
    assert h.hexdigest() == V2_VSS_SHA256, "the recorded VSS goal-mouth scrums changed: model v2 of DESIGN.md 4 is frozen"


@pytest.mark.parametrize("k", [0, 1])
def test_model_v2_vss_scrums_are_reproduced_and_sit_at_the_goal(oracle_mod, k):
    """the oracle reproduces the recorded six-robot scrums (400 steps each, sampled every 20th), and the recordings ARE what they are
    for: robots held by the goal line's wall and by the goal box's walls, robots past the posts' chords, and a pile that stays a pile
    of discs (overlap below 4 mm in every sample) — held axes, chord posts and the wall clamp all take part"""
    z = np.load(V2_VSS_GOLDEN)
    e = oracle_mod.OracleEnv(0, 0, 3, 3, 25, "f64")
    e.set_state_full(z[f"vss{k}_reset_state"])
    cmds, want = z[f"vss{k}_cmds"].astype(np.float64), z[f"vss{k}_states"]
    f = e.field_params()
    hl, ghw, gd, r = f[0] / 2, f[4] / 2, f[5], f[14]
    at_line = in_goal = in_corner = 0
    for t in range(len(cmds)):
        e.step(cmds[t])
        if t % 20 == 19:
            got = e.get_state_full()
            assert np.allclose(got, want[t // 20], rtol=0, atol=1e-12), (k, t)
            x, y = got[5::6][:6], got[6::6][:6]
            d = np.hypot(x[:, None] - x[None], y[:, None] - y[None]) + 9.0 * np.eye(6)
            assert 0.075 - d.min() < 0.004
            ax, ay = np.abs(x), np.abs(y)
            at_line += int(((np.abs(ax - (hl - r)) < 1e-9) & (ay >= ghw)).any())
            in_goal += int((ax > hl).any())
            in_corner += int(((ax > hl - r + 1e-6) & (ax <= hl) & (ay < ghw)).any())
            assert (ax[(ax <= hl) & (ay < ghw)] + ay[(ax <= hl) & (ay < ghw)] <= hl + ghw - r + 1e-9).all()   # the chord holds
            assert (ay[ax > hl] <= ghw - r + 1e-9).all() and (ax <= hl + gd - r + 1e-9).all()
    assert at_line >= 3 and in_goal >= 10 and in_corner >= 3, (at_line, in_goal, in_corner)
