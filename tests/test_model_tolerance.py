"""What CAN be pinned about the physics without rc-robosim:

* the STATED fp32 tolerance — BASELINE.json's north star asks for results that match the CPU path
  "to a stated fp32 tolerance on positions/velocities"; the CPU path's boundary type is float64
  (rsoccer_gym/Simulators/rsim.py:105 returns f64).  The float32 model (what the GPU computes,
  bit for bit) is compared here with the float64 instantiation of the same model;
* the behaviour the reference's task code REQUIRES of its simulator: the literals
  the tasks hard-code (possession at 0.1 m, hold at 0.115 m, static robots below 0.05 m/s,
  infrared <=> ball at the kicker face) must hold in this model or those tasks cannot work.
"""
import math

import numpy as np
import pytest

# The stated tolerance (also in include/rsx.h and DESIGN.md 2): over 1 s of simulated time
# (40 steps of 25 ms = 200 sub-steps) from the same f32-representable state under the same commands
POS_TOL_M = 1e-4       # |x_f32 - x_f64|, every body, metres
VEL_TOL_MS = 1e-3      # |v_f32 - v_f64|, every body, m/s
HORIZON = 40
# crowded 22-robot scrum (BASELINE.json configs[3], "worst-case all-pairs contacts"): a contact onset
# can fall into neighbouring 5 ms sub-steps in the two precisions, after which that env's bodies
# separate at the relative impact speed; >= 95 % of the envs still meet the tight tolerance and
# every env stays within
CROWD_POS_TOL_M, CROWD_VEL_TOL_MS, CROWD_TIGHT_FRACTION = 1e-2, 1e-1, 0.95

CASES = {"vss3v3": (0, 0, 3, 3, 1), "ssl1v6": (1, 2, 1, 6, 2), "ssl1v4": (1, 2, 1, 4, 3),
         "ssl1v1": (1, 2, 1, 1, 4), "ssl2v0": (1, 2, 2, 0, 5), "ssl11v11": (1, 1, 11, 11, 0)}


def _commands(rng, kind, N, actuators=True):
    if kind == 0:
        return rng.uniform(-40, 40, (N, 2))
    cm = rng.uniform(-1, 1, (N, 8))
    cm[:, 0] = 0; cm[:, 1:3] *= 2.0; cm[:, 3] *= 6.0; cm[:, 4] = 0
    cm[:, 5] = (cm[:, 5] > 0.8) * 4.0; cm[:, 6] = (cm[:, 6] > 0.9) * 2.0; cm[:, 7] = cm[:, 7] > 0
    if not actuators:
        cm[:, 5:8] = 0
    return cm


@pytest.mark.parametrize("name", list(CASES))
def test_f32_model_tracks_f64_within_the_stated_tolerance(oracle_mod, name):
    """40 envs per configuration, started from the task's own reset placements (or a crowded line-up
    for 11v11), random commands incl. kicks, chips and dribbling: contacts happen, and the two
    precisions stay within POS_TOL_M / VEL_TOL_MS over HORIZON steps.  The tolerance is a statement
    about the continuous dynamics and the contacts; kicker and dribbler are threshold devices (they
    fire when the ball is within 2.5 cm of the face), so in the crowded 22-robot case — where the
    ball is always next to some kicker — they are left off: one flipped infrared decision is a 4 m/s
    difference in either precision of ANY simulator."""
    O = oracle_mod
    kind, ft, nb, ny, task = CASES[name]
    N, RS = nb + ny, 6 if kind == 0 else 11
    rng = np.random.default_rng(11)
    ip = [0, 1] + [5 + RS * k + j for k in range(N) for j in (0, 1)]
    iv = [3, 4] + [5 + RS * k + j for k in range(N) for j in (3, 4)]
    per_env = []
    contacts = 0
    for i in range(40):
        worst_p = worst_v = 0.0
        a, b = O.OracleEnv(kind, ft, nb, ny, 25, "f32"), O.OracleEnv(kind, ft, nb, ny, 25, "f64")
        if task:
            a.task_attach(task, 5, i, 10 ** 6)
            a.task_reset()
        else:
            pts = [(0.3 * (k % 6 - 2.5) + rng.uniform(-.05, .05), 0.3 * (k // 6 - 1.5) + rng.uniform(-.05, .05)) for k in range(N)]
            pose = np.array([[p[0], p[1], rng.uniform(-180, 180)] for p in pts])
            a.reset([0.1, 0.9, 1.0, -2.0], pose[:nb], pose[nb:])
        start = a.get_state_full()          # f32 values; the resting height is the one constant that is
        start[2] = 0.0215                   # not f32-representable: give the f64 env ITS resting height
        b.set_state_full(start)
        for _ in range(HORIZON):
            cm = _commands(rng, kind, N, actuators=task != 0)
            a.step(cm); b.step(cm)
            sa, sb = a.get_state(), b.get_state()
            worst_p = max(worst_p, np.abs(sa[ip] - sb[ip]).max())
            worst_v = max(worst_v, np.abs(sa[iv] - sb[iv]).max())
            P = sb[5:].reshape(N, RS)[:, :2]
            if N > 1:
                D = np.hypot(*(P[:, None] - P[None]).transpose(2, 0, 1)); np.fill_diagonal(D, 9.0)
                contacts += D.min() < 2 * b.field_params()[14] + 1e-3
            contacts += np.hypot(P[:, 0] - sb[0], P[:, 1] - sb[1]).min() < b.field_params()[14] + 0.0215 + 1e-3
        per_env.append((worst_p, worst_v))
    per_env = np.array(per_env)
    tight = (per_env[:, 0] <= POS_TOL_M) & (per_env[:, 1] <= VEL_TOL_MS)
    if task:
        assert tight.all(), per_env.max(0)
    else:
        assert tight.mean() >= CROWD_TIGHT_FRACTION, tight.mean()
        assert per_env[:, 0].max() <= CROWD_POS_TOL_M and per_env[:, 1].max() <= CROWD_VEL_TOL_MS, per_env.max(0)
    assert contacts > 0          # the sample is not contact-free


def _ssl(O, prec, ball, blue, yellow=(), ft=2):
    e = O.OracleEnv(1, ft, len(blue), len(yellow), 25, prec)
    e.reset(np.array(ball, float), np.array(blue, float), np.array(yellow, float).reshape(-1, 3))
    return e


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_static_robots_stay_below_the_collision_threshold_unless_hit(oracle_mod, prec):
    """dribbling.py:143-145 / contested_possession.py:166-169 end the episode when an undriven robot
    moves faster than 0.05 m/s (0.1 m/s): idle robots must be exactly at rest, and a real hit must
    be visible above the threshold."""
    e = _ssl(oracle_mod, prec, [2, 1.5, 0, 0], [[0, 0, 180.0]], [[-0.5 * k, 0, 180.0] for k in range(1, 5)])
    for _ in range(200):
        c = np.zeros((5, 8)); c[0, 1:4] = [0.0, 0.8, 0.0]     # the agent moves sideways, away from the obstacles
        e.step(c)
        y = e.get_state()[5 + 11:].reshape(4, 11)
        assert np.all(y[:, 3:6] == 0.0) and np.all(np.abs(y[:, 0] - [-0.5, -1.0, -1.5, -2.0]) == 0)
    e = _ssl(oracle_mod, prec, [2, 1.5, 0, 0], [[0, 0, 180.0]], [[-0.5 * k, 0, 180.0] for k in range(1, 5)])
    hit = False
    for _ in range(40):
        c = np.zeros((5, 8)); c[0, 1] = 1.5                    # heading 180 deg: local +x is global -x, into obstacle 1
        e.step(c)
        y1 = e.get_state()[5 + 11:5 + 22]
        hit |= abs(y1[3]) > 0.05 or abs(y1[4]) > 0.05
    assert hit


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_possession_and_hold_distances_of_the_tasks(oracle_mod, prec):
    """dribbling.py:193-195 places the ball 0.1 m in front of the robot and expects the dribbler to
    keep it; pass_endurance.py:169-174 places it at 0.115 m and expects the shooter to hold it.
    Both distances must switch the infrared sensor on with the dribbler running, and the ball must
    follow the robot."""
    for dist in (0.1, 0.115):
        e = _ssl(oracle_mod, prec, [dist, 0, 0, 0], [[0, 0, 0.0]])
        c = np.zeros((1, 8)); c[0, 7] = 1.0
        e.step(c)
        assert e.get_state()[5 + 6] == 1.0                       # infrared
        c[0, 1] = -0.5                                          # drive backwards: the ball must come along
        for _ in range(40):
            e.step(c)
        st = e.get_state()
        assert st[5 + 6] == 1.0 and st[5] < -0.3
        assert abs(math.hypot(st[0] - st[5], st[1] - st[6]) - 0.0945) < 0.01   # held at the kicker face
    # without the dribbler the ball stays where it was
    e = _ssl(oracle_mod, prec, [0.1, 0, 0, 0], [[0, 0, 0.0]])
    c = np.zeros((1, 8)); c[0, 1] = -0.5
    for _ in range(40):
        e.step(c)
    assert abs(e.get_state()[0] - 0.1) < 1e-6 and e.get_state()[5 + 6] == 0.0


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_infrared_means_ball_at_the_kicker_face(oracle_mod, prec):
    """pass_endurance.py:134 ends the episode with success when the receiver's infrared is on: the
    flag must be on exactly when the ball is in front of the kicker (inside its width, within
    2.5 cm of the face), and off beside, behind or away from the robot."""
    f = _ssl(oracle_mod, prec, [2, 1, 0, 0], [[0, 0, 0.0]]).field_params()
    face = f[7] + f[6]              # centre -> ball centre when touching the kicker plate
    half_w = f[9] / 2
    for (bx, by, want) in [(face + 0.01, 0.0, 1), (face + 0.024, 0.0, 1), (face + 0.03, 0.0, 0),
                           (face + 0.01, half_w - 0.005, 1), (face + 0.01, half_w + 0.01, 0),
                           (-(face + 0.03), 0.0, 0), (0.0, 0.0945 + 0.03, 0), (1.0, 0.0, 0)]:
        for th in (0.0, 90.0, -135.0):
            c, s = math.cos(math.radians(th)), math.sin(math.radians(th))
            e = _ssl(oracle_mod, prec, [bx * c - by * s, bx * s + by * c, 0, 0], [[0, 0, th]])
            e.step(np.zeros((1, 8)))
            assert e.get_state()[5 + 6] == want, (bx, by, th)
    # a chipped ball above the robot does not trip it
    e = _ssl(oracle_mod, prec, [face + 0.01, 0, 0, 0], [[0, 0, 0.0]])
    st = e.get_state_full(); st[2] = 0.0215 + 0.3; e.set_state_full(st)
    e.step(np.zeros((1, 8)))
    assert e.get_state()[5 + 6] == 0.0
