"""More independent pins for the 2-D step model: closed-form answers (impulse exchange, ballistics, bounce height,
acceleration limits, the friction cone) and the symmetries of the field (mirror, half turn).  None of these
compares the model with itself: each expectation comes from the constants in DESIGN.md section 4 and plain
mechanics, or from the geometry of the pitch.  Same three backends as tests/test_physics_model.py."""
import math

import numpy as np
import pytest

from test_physics_model import BACKENDS, _cmd, _make, _ssl, _vss

G = 9.81
SSL = dict(m_robot=2.2, m_ball=0.046, e_rb=0.2, r_robot=0.09, r_ball=0.0215, a_lin=5.0, a_ang=50.0, mu_g=0.4, e_ground=0.5,
           mu_wb=0.3, e_wb=0.5)
VSS = dict(m_robot=0.18, m_ball=0.046, e_rb=0.3, r_robot=0.0375, r_ball=0.0215, mu_g=0.3, e_wb=0.6, mu_wb=0.3)


def _tol(backend, t64, t32):
    return t64 if backend == "f64" else t32


@pytest.mark.parametrize("backend", BACKENDS)
def test_ball_bounces_off_a_resting_robot_with_the_two_body_impulse(backend, oracle_mod):
    """a ball hitting the BACK of a resting SSL robot head-on (body circle, not the kicker): relative normal speed
    reversed with e_rb, shared by inverse mass: v' = v (1 - (1 + e) * (1/m_b) / (1/m_r + 1/m_b))"""
    for model, make, x0 in ((SSL, lambda b: _ssl(backend, b, [0.0, 0.0, 0.0]), -0.4),
                            (VSS, lambda b: _vss(backend, b, [0.0, 0.0, 0.0]), -0.25)):
        s = make([x0, 0.0, 2.0, 0.0])     # robot at the origin facing +x: the ball comes from behind
        n_cmd = (1, 8) if model is SSL else (6, 2)
        v_in = v_out = None
        for _ in range(30):
            before = s.get_state()[3]
            s.step(np.zeros(n_cmd))
            after = s.get_state()[3]
            if before > 0 and after < 0:
                v_in, v_out = before, after
                break
        assert v_in is not None, "the ball never bounced"
        w_b = (1 / model["m_ball"]) / (1 / model["m_robot"] + 1 / model["m_ball"])
        want = 1.0 - (1.0 + model["e_rb"]) * w_b
        # v_in is the speed one step before the hit: rolling resistance takes mu_g * dt off it, at most
        assert abs(v_out / v_in - want) < 0.02, (v_out / v_in, want)
        assert abs(s.get_state()[4]) < 1e-6          # head-on: nothing sideways


@pytest.mark.parametrize("backend", BACKENDS)
def test_chip_kick_is_a_parabola_and_the_bounce_loses_the_ground_restitution(backend, oracle_mod):
    vx, vz = 2.0, 3.0
    s = _ssl(backend, [0.1, 0.0, 0, 0], [0, 0, 0.0], ft=1)     # big field: no wall in the way
    s.step(_cmd(1, 8, {0: [0, 0, 0, 0, 0, vx, vz, 0]}))
    z = [s.get_state()[2] - SSL["r_ball"]]
    x = [s.get_state()[0]]
    for _ in range(80):
        s.step(np.zeros((1, 8)))
        st = s.get_state()
        z.append(st[2] - SSL["r_ball"]); x.append(st[0])
    z = np.array(z)
    apex1 = z.max()
    assert abs(apex1 - vz ** 2 / (2 * G)) < 0.05 * vz ** 2 / (2 * G)          # h = vz^2 / 2g
    landed = int(np.argmax((z[1:] == 0.0) | (np.diff(z) > 0) & (z[:-1] < 0.02) & (np.arange(len(z) - 1) > 5))) + 1
    t_flight = landed * 0.025
    assert abs(t_flight - 2 * vz / G) < 0.06                                   # T = 2 vz / g (one step of slack)
    assert abs((x[landed] - x[0]) - vx * t_flight) < 0.06                      # no drag in the air
    apex2 = z[landed + 1:landed + 1 + int(2 * vz * SSL["e_ground"] / G / 0.025) + 2].max()
    assert abs(apex2 / apex1 - SSL["e_ground"] ** 2) < 0.08                    # second apex = e^2 x first


@pytest.mark.parametrize("backend", BACKENDS)
def test_holonomic_robot_ramps_at_its_acceleration_limits(backend, oracle_mod):
    s = _ssl(backend, [3.0, 2.0, 0, 0], [-3.0, 0.0, 0.0], ft=1)
    v = []
    for _ in range(20):
        s.step(_cmd(1, 8, {0: [0, 2.0, 0.0, 0.0]}))
        v.append(s.get_state()[8])
    v = np.array(v)
    ramp = v[v < 1.9]
    assert len(ramp) >= 8 and np.allclose(np.diff(ramp), SSL["a_lin"] * 0.025, atol=2e-4)   # dv = a_lin dt
    assert abs(v[-1] - 2.0) < 1e-4                                                          # and holds the target
    s = _ssl(backend, [3.0, 2.0, 0, 0], [0.0, 0.0, 0.0], ft=1)
    w = []
    for _ in range(6):
        s.step(_cmd(1, 8, {0: [0, 0.0, 0.0, 8.0]}))
        w.append(math.radians(s.get_state()[10]))
    w = np.array(w)
    assert np.allclose(np.diff(w[w < 7.0]), SSL["a_ang"] * 0.025, atol=1e-3)                 # dw = a_ang dt


@pytest.mark.parametrize("backend", BACKENDS)
def test_oblique_wall_hit_stays_inside_the_friction_cone(backend, oracle_mod):
    """ball against the VSS side wall at 45 degrees: normal speed reversed with e_wb, tangential speed reduced by
    at most mu (1 + e) |v_n| (Coulomb)"""
    s = _vss(backend, [0.0, 0.45, 1.0, 1.0], [0.0, -0.5, 0.0])
    for _ in range(12):
        b = s.get_state()[[3, 4]].copy()
        s.step(np.zeros((6, 2)))
        a = s.get_state()[[3, 4]]
        if b[1] > 0 and a[1] < 0:
            vn, dvt = b[1], b[0] - a[0]
            assert abs(-a[1] / vn - VSS["e_wb"]) < 0.03
            assert 0.0 <= dvt <= VSS["mu_wb"] * (1 + VSS["e_wb"]) * vn + 0.3 * 0.025 + 1e-3    # + rolling resistance of the step
            return
    raise AssertionError("the ball never reached the wall")


def _mirror_x(ball, robots):
    b = np.array(ball, float) * [-1, 1, -1, 1]
    r = np.array(robots, float).copy()
    r[:, 0] *= -1; r[:, 2] = 180.0 - r[:, 2]
    return b, r


def _half_turn(ball, robots):
    b = -np.array(ball, float)
    r = np.array(robots, float).copy()
    r[:, :2] *= -1; r[:, 2] = r[:, 2] + 180.0
    return b, r


def _unwrap_deg(a):
    return (np.asarray(a) + 180.0) % 360.0 - 180.0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("sym", ["mirror-x", "half-turn"])
def test_the_pitch_is_symmetric_vss(backend, sym, oracle_mod):
    """The same scene reflected in the half-way line (blue and yellow keep their sides' geometry: both halves of the
    pitch are equal) or turned by 180 degrees evolves into the reflected / turned scene: walls, goal mouths,
    corners, contacts and friction have no preferred direction.  Differential drive: a reflection swaps the wheels."""
    rng = np.random.default_rng(11)
    ball = [0.05, -0.1, 0.9, 0.7]
    robots = np.column_stack([rng.uniform(-0.6, 0.6, 6), rng.uniform(-0.5, 0.5, 6), rng.uniform(-180, 180, 6)])
    robots[0] = [-0.05, -0.2, 40.0]      # one of them right next to the ball
    cmds = rng.uniform(-25, 25, (60, 6, 2))
    tf = _mirror_x if sym == "mirror-x" else _half_turn
    a = _make(backend, 0, 0, 3, 3)
    a.reset(np.array(ball, float), robots[:3], robots[3:])
    b2, r2 = tf(ball, robots)
    b = _make(backend, 0, 0, 3, 3)
    b.reset(b2, r2[:3], r2[3:])
    tol = _tol(backend, 1e-7, 2e-3)
    for t in range(60):
        a.step(cmds[t])
        b.step(cmds[t][:, ::-1] if sym == "mirror-x" else cmds[t])
        sa, sb = a.get_state(), b.get_state()
        eb, er = tf(sa[[0, 1, 3, 4]], sa[5:].reshape(6, 6)[:, :3])
        assert np.allclose(sb[[0, 1, 3, 4]], eb, atol=tol), (t, sb[[0, 1, 3, 4]], eb)
        gr = sb[5:].reshape(6, 6)[:, :3]
        assert np.allclose(gr[:, :2], er[:, :2], atol=tol), t
        assert np.allclose(_unwrap_deg(gr[:, 2] - er[:, 2]), 0.0, atol=tol * 100), t
    assert np.hypot(*(a.get_state()[[0, 1]] - np.array(ball[:2]))) > 0.2     # something did happen


@pytest.mark.parametrize("backend", BACKENDS)
def test_the_pitch_is_symmetric_ssl(backend, oracle_mod):
    """half turn of an SSL scene with kicks, the dribbler and robot-robot hits (local velocity commands turn with
    the robot, so the command stream is unchanged)"""
    rng = np.random.default_rng(5)
    ball = [0.105, 0.0, 0.0, 0.0]
    robots = np.array([[0.0, 0.0, 0.0], [0.8, 0.1, 170.0], [0.5, -0.6, 90.0], [-0.9, 0.7, -30.0]])
    cmds = np.zeros((50, 4, 8))
    cmds[:, :, 1:4] = rng.uniform(-1, 1, (50, 4, 3)) * [1.5, 1.5, 4.0]
    cmds[5:, 0, 5] = 3.0; cmds[:, 0, 7] = 1.0; cmds[20:, 1, 5] = 2.0
    a = _make(backend, 1, 2, 1, 3)
    a.reset(np.array(ball, float), robots[:1], robots[1:])
    b2, r2 = _half_turn(ball, robots)
    b = _make(backend, 1, 2, 1, 3)
    b.reset(b2, r2[:1], r2[1:])
    tol = _tol(backend, 1e-7, 2e-3)
    for t in range(50):
        a.step(cmds[t]); b.step(cmds[t])
        sa, sb = a.get_state(), b.get_state()
        assert np.allclose(sb[[0, 1, 3, 4]], -sa[[0, 1, 3, 4]], atol=tol), t
        ra, rb = sa[5:].reshape(4, 11), sb[5:].reshape(4, 11)
        assert np.allclose(rb[:, [0, 1, 3, 4]], -ra[:, [0, 1, 3, 4]], atol=tol), t
        assert np.allclose(rb[:, 5:], ra[:, 5:], atol=tol * 1e3), t        # yaw rate, infrared, wheel speeds: unchanged
    assert abs(a.get_state()[0]) > 0.3       # the ball was kicked away
