"""Known-answer and invariant tests of the 2-D step model (DESIGN.md "Physics model").

The model is this project's own specification (the reference's physics lives in rc-robosim,
which is absent); these tests are what defines acceptable behaviour, derived from what the
reference's task code expects of a simulator.  They run against the CPU oracle
in both precisions here, and — marked gpu — against the HIP engine through the
robosim-compatible classes.
"""
import math

import numpy as np
import pytest


def _make(backend, kind, ft, nb, ny, ts=25):
    line = lambda n, s: [[s * 0.2 * i, 0, 0] for i in range(1, n + 1)]
    if backend == "hip":
        from rsoccer_amd import robosim
        cls = robosim.VSS if kind == 0 else robosim.SSL
        return cls(ft, nb, ny, ts, [0, 0, 0, 0], line(nb, -1), line(ny, 1))
    import fake_robosim
    fake_robosim.arm()
    fake_robosim.PREC = backend
    cls = fake_robosim.VSS if kind == 0 else fake_robosim.SSL
    try:
        return cls(ft, nb, ny, ts, [0, 0, 0, 0], line(nb, -1), line(ny, 1))
    finally:
        fake_robosim.PREC = "f64"


BACKENDS = [pytest.param("f64", id="oracle-f64"), pytest.param("f32", id="oracle-f32"),
            pytest.param("hip", id="hip", marks=pytest.mark.gpu)]
FAR = [[5.0 + i, 5.0, 0.0] for i in range(8)]  # parking spots outside every field (walls clamp them)


def _vss(backend, ball, blue0, others=None):
    s = _make(backend, 0, 0, 3, 3)
    blue = [blue0, [-0.6, 0.5, 0.0], [-0.6, -0.5, 0.0]]
    yel = [[0.6, 0.5, 0.0], [0.6, 0.0, 0.0], [0.6, -0.5, 0.0]]
    s.reset(np.array(ball, float), np.array(blue, float), np.array(yel, float))
    return s


def _ssl(backend, ball, blue0, ft=2, ny=0, yellow=None):
    s = _make(backend, 1, ft, 1, ny)
    s.reset(np.array(ball, float), np.array([blue0], float), np.array(yellow if yellow is not None else np.zeros((0, 3)), float))
    return s


def _cmd(n, c, rows):
    a = np.zeros((n, c))
    for i, r in rows.items():
        a[i, :len(r)] = r
    return a


@pytest.mark.parametrize("backend", BACKENDS)
def test_state_layout_and_field(backend, oracle_mod):
    s = _vss(backend, [0.1, 0.2, 0.3, -0.4], [-0.3, 0.1, 45.0])
    st = s.get_state()
    assert st.shape == (41,) and st.dtype == np.float64
    f = s.get_field_params()
    assert list(f)[:3] == ["length", "width", "penalty_length"] and len(f) == 17
    assert np.allclose(st[:5], [0.1, 0.2, f["ball_radius"], 0.3, -0.4], atol=1e-7)
    assert np.allclose(st[5:11], [-0.3, 0.1, 45.0, 0, 0, 0], atol=1e-5)
    s2 = _ssl(backend, [1, 0, 0, 0], [0, 0, 0], ny=2, yellow=[[2, 1, 10], [2, -1, 20]])
    assert s2.get_state().shape == (5 + 11 * 3,)


@pytest.mark.parametrize("backend", BACKENDS)
def test_differential_drive_straight_and_arc(backend, oracle_mod):
    f = _vss(backend, [0.6, 0.55, 0, 0], [0, 0, 0]).get_field_params()
    rw, b = f["rbt_wheel_radius"], 0.04
    # equal wheels: straight line along the heading, speed -> r_w * w
    s = _vss(backend, [0.6, 0.55, 0, 0], [-0.5, 0.0, 0.0])
    for _ in range(20):
        s.step(_cmd(6, 2, {0: [20.0, 20.0]}))
    st = s.get_state()
    assert abs(st[8] - rw * 20.0) < 1e-5 and abs(st[9]) < 1e-6 and abs(st[6]) < 1e-6 and abs(st[10]) < 1e-4
    assert -0.5 < st[5] < 0.2
    # unequal wheels: omega -> r_w (wr - wl) / (2 b), turning radius v / omega
    s = _vss(backend, [0.6, 0.55, 0, 0], [0.0, 0.0, 0.0])
    for _ in range(12):
        s.step(_cmd(6, 2, {0: [10.0, 14.0]}))
    st = s.get_state()
    om = math.radians(st[10]); v = math.hypot(st[8], st[9])
    assert abs(om - rw * 4.0 / (2 * b)) < 1e-3
    assert abs(v / om - (rw * 12.0) / (rw * 4.0 / (2 * b))) < 2e-3
    # wheel speeds saturate at the motor limit
    s = _vss(backend, [0.6, 0.55, 0, 0], [-0.5, 0.0, 0.0])
    for _ in range(30):
        s.step(_cmd(6, 2, {0: [1e4, 1e4]}))
    wmax = f["rbt_motor_max_rpm"] / 60 * 2 * math.pi
    assert abs(s.get_state()[8] - rw * wmax) < 1e-4 or abs(s.get_state()[5]) > 0.6  # reached top speed (or the wall)


@pytest.mark.parametrize("backend", BACKENDS)
def test_idle_robots_stay_at_rest(backend, oracle_mod):
    """contested_possession.py:166-169 / dribbling.py:143-145 read any yellow speed > 0.05 m/s as
    a collision: untouched robots with zero commands must not creep."""
    s = _vss(backend, [0.0, 0.3, 0, 0], [0, -0.3, 30.0])
    before = s.get_state()
    for _ in range(40):
        s.step(np.zeros((6, 2)))
    after = s.get_state()
    assert np.array_equal(after[5:], before[5:])
    y = [[1.0, 0.5, 0], [2.0, -0.5, 90.0]]
    s = _ssl(backend, [0.5, 1.5, 0, 0], [0, 0, 0], ny=2, yellow=y)
    before = s.get_state()
    for _ in range(40):
        s.step(np.zeros((3, 8)))
    assert np.array_equal(s.get_state()[5:], before[5:])


@pytest.mark.parametrize("backend", BACKENDS)
def test_ball_rolls_to_a_stop(backend, oracle_mod):
    s = _vss(backend, [-0.5, 0.0, 0.6, 0.0], [0, 0.5, 0])
    xs = []
    for _ in range(120):
        s.step(np.zeros((6, 2)))
        xs.append(s.get_state()[[0, 3]].copy())
    xs = np.array(xs)
    assert xs[-1, 1] == 0.0                                  # stopped exactly
    decel = (xs[0, 1] - xs[10, 1]) / (10 * 0.025)
    assert abs(decel - 0.3) < 1e-3                            # constant rolling deceleration
    assert abs((xs[-1, 0] + 0.5) - 0.6 ** 2 / (2 * 0.3)) < 0.01  # stopping distance v^2 / (2 a)
    assert np.all(np.diff(xs[:, 1]) <= 1e-9)


@pytest.mark.parametrize("backend", BACKENDS)
def test_walls_and_goal_mouth(backend, oracle_mod):
    f = _vss(backend, [0, 0, 0, 0], [0, 0.5, 0]).get_field_params()
    L, W, gw, gd, rb = f["length"], f["width"], f["goal_width"], f["goal_depth"], f["ball_radius"]
    # side wall: the ball bounces back with the wall's restitution and never leaves the field
    s = _vss(backend, [0.0, 0.55, 0.0, 1.0], [0, -0.5, 0])
    vy = []
    for _ in range(8):
        s.step(np.zeros((6, 2)))
        st = s.get_state()
        assert abs(st[1]) <= W / 2 - rb + 1e-6
        vy.append(st[4])
    assert min(vy) < -0.5 and abs(min(vy) / 1.0 + 0.6) < 0.03
    # end wall outside the goal mouth: bounce
    s = _vss(backend, [0.6, 0.4, 1.5, 0.0], [0, -0.5, 0])
    for _ in range(10):
        s.step(np.zeros((6, 2)))
        assert s.get_state()[0] <= L / 2 - rb + 1e-6
    assert s.get_state()[3] < 0
    # through the goal mouth: the ball crosses x = L/2 (what vss_gym.py:161 calls a goal) and is
    # stopped by the back wall of the goal
    s = _vss(backend, [0.6, 0.05, 1.5, 0.0], [0, -0.5, 0])
    crossed = False
    for _ in range(12):
        s.step(np.zeros((6, 2)))
        st = s.get_state()
        crossed |= st[0] > L / 2
        assert st[0] <= L / 2 + gd - rb + 1e-6 and abs(st[1]) <= gw / 2 - rb + 1e-6 or st[0] <= L / 2
    assert crossed
    # a robot driven into a corner stays inside
    s = _vss(backend, [0, 0, 0, 0], [0.6, 0.5, 40.0])
    for _ in range(40):
        s.step(_cmd(6, 2, {0: [40.0, 40.0]}))
    st = s.get_state()
    assert st[5] <= L / 2 - f["rbt_radius"] + 1e-6 and st[6] <= W / 2 - f["rbt_radius"] + 1e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_robot_ball_collision(backend, oracle_mod):
    """a robot pushing into a resting ball sends it away along the line of centres; momentum of
    the pair only changes through the motor (no energy from nowhere)."""
    s = _vss(backend, [0.0, 0.0, 0, 0], [-0.2, 0.0, 0.0])
    hit = False
    for _ in range(30):
        s.step(_cmd(6, 2, {0: [30.0, 30.0]}))
        st = s.get_state()
        gap = math.hypot(st[0] - st[5], st[1] - st[6])
        assert gap > 0.0375 + 0.0215 - 3e-3          # never deeply interpenetrating
        hit |= st[3] > 0.1
    st = s.get_state()
    assert hit and st[3] > 0 and abs(st[4]) < 1e-5 and st[0] > 0.05
    # robot - robot: two robots driven against each other stop each other, no tunnelling
    s = _make(backend, 0, 0, 3, 3)
    s.reset(np.array([0, 0.5, 0, 0.0]), np.array([[-0.1, 0, 0.0], [-0.6, 0.5, 0], [-0.6, -0.5, 0]]),
            np.array([[0.1, 0, 180.0], [0.6, 0.5, 0], [0.6, -0.5, 0]]))
    for _ in range(40):
        s.step(_cmd(6, 2, {0: [30.0, 30.0], 3: [30.0, 30.0]}))
        st = s.get_state()
        assert st[5 + 18] - st[5] > 2 * 0.0375 - 4e-3
    assert st[5] < 0 < st[5 + 18]


@pytest.mark.parametrize("backend", BACKENDS)
def test_holonomic_commands(backend, oracle_mod):
    # local (v_x, v_y, v_theta) command: heading 90 deg, local +x is global +y
    s = _ssl(backend, [2.0, 1.5, 0, 0], [0, 0, 90.0])
    for _ in range(30):
        s.step(_cmd(1, 8, {0: [0, 1.0, 0.0, 0.0]}))
    st = s.get_state()
    assert abs(st[9] - 1.0) < 1e-4 and abs(st[8]) < 1e-4 and st[6] > 0.3 and abs(st[5]) < 1e-3
    # pure rotation at 2 rad/s
    s = _ssl(backend, [2.0, 1.5, 0, 0], [0, 0, 0.0])
    for _ in range(20):
        s.step(_cmd(1, 8, {0: [0, 0.0, 0.0, 2.0]}))
    assert abs(math.radians(s.get_state()[10]) - 2.0) < 1e-4
    # the reported wheel speeds (static_defenders.py:311-322 reads them) reproduce the motion
    # when fed back as a wheel-speed command
    s = _ssl(backend, [2.0, 1.5, 0, 0], [0, 0, 30.0])
    for _ in range(25):
        s.step(_cmd(1, 8, {0: [0, 0.8, -0.4, 1.0]}))
    st = s.get_state()
    wheels = st[12:16]
    s2 = _ssl(backend, [2.0, 1.5, 0, 0], [0, 0, 30.0])
    for _ in range(25):
        s2.step(_cmd(1, 8, {0: [1, *wheels]}))
    st2 = s2.get_state()
    # (the heading differs by the acceleration transient, so compare robot-frame quantities)
    assert abs(math.hypot(*st2[8:10]) - math.hypot(*st[8:10])) < 1e-4 and abs(st2[10] - st[10]) < 1e-3
    assert np.allclose(st2[12:16], wheels, atol=0.3)   # within one sub-step of heading change (omega * h)
    # wheel speeds saturate at 160 rad/s (static_defenders.py:71)
    s = _ssl(backend, [2.0, 1.5, 0, 0], [-2.0, 0, 0.0])
    for _ in range(60):
        s.step(_cmd(1, 8, {0: [0, 50.0, 0.0, 0.0]}))
    assert np.max(np.abs(s.get_state()[12:16])) <= 160.0 + 1e-3


@pytest.mark.parametrize("backend", BACKENDS)
def test_infrared_kick_and_dribbler(backend, oracle_mod):
    # a ball 0.1 m in front of the robot centre is "possessed" (dribbling.py:193-195)
    s = _ssl(backend, [0.1, 0.0, 0, 0], [0, 0, 0.0])
    s.step(np.zeros((1, 8)))
    assert s.get_state()[11] == 1.0
    s = _ssl(backend, [0.3, 0.0, 0, 0], [0, 0, 0.0])
    s.step(np.zeros((1, 8)))
    assert s.get_state()[11] == 0.0
    s = _ssl(backend, [0.0, 0.1, 0, 0], [0, 0, 0.0])  # beside, not in front
    s.step(np.zeros((1, 8)))
    assert s.get_state()[11] == 0.0
    # kick: the ball leaves along the heading at kick_v_x (static_defenders.py:78,125)
    s = _ssl(backend, [0.1 * math.cos(math.radians(30)), 0.1 * math.sin(math.radians(30)), 0, 0], [0, 0, 30.0])
    s.step(_cmd(1, 8, {0: [0, 0, 0, 0, 0, 5.0, 0.0, 0]}))
    st = s.get_state()
    sp = math.hypot(st[3], st[4])
    assert 4.8 < sp <= 5.0 + 1e-5 and abs(math.degrees(math.atan2(st[4], st[3])) - 30.0) < 0.5
    # no kick without infrared
    s = _ssl(backend, [0.5, 0.0, 0, 0], [0, 0, 0.0])
    s.step(_cmd(1, 8, {0: [0, 0, 0, 0, 0, 5.0, 0.0, 0]}))
    assert np.all(s.get_state()[3:5] == 0)
    # dribbler: the ball is carried by a moving, turning robot
    s = _ssl(backend, [0.1, 0.0, 0, 0], [0, 0, 0.0])
    for _ in range(60):
        s.step(_cmd(1, 8, {0: [0, 0.5, 0.0, 1.0, 0, 0, 0, 1]}))
        st = s.get_state()
    th = math.radians(st[7])
    lx = (st[0] - st[5]) * math.cos(th) + (st[1] - st[6]) * math.sin(th)
    ly = -(st[0] - st[5]) * math.sin(th) + (st[1] - st[6]) * math.cos(th)
    assert st[11] == 1.0 and abs(lx - (0.073 + 0.0215)) < 0.012 and abs(ly) < 0.02
    # same drive without dribbler: the ball is left behind / pushed away
    s = _ssl(backend, [0.1, 0.0, 0, 0], [0, 0, 0.0])
    for _ in range(60):
        s.step(_cmd(1, 8, {0: [0, 0.5, 0.0, 1.0, 0, 0, 0, 0]}))
    st = s.get_state()
    th = math.radians(st[7])
    ly = -(st[0] - st[5]) * math.sin(th) + (st[1] - st[6]) * math.cos(th)
    assert st[11] == 0.0 or abs(ly) > 0.02


@pytest.mark.parametrize("backend", BACKENDS)
def test_chip_kick_flies_over_a_robot(backend, oracle_mod):
    y = [[0.6, 0.0, 0.0]]
    s = _ssl(backend, [0.1, 0.0, 0, 0], [0, 0, 0.0], ny=1, yellow=y)
    s.step(_cmd(2, 8, {0: [0, 0, 0, 0, 0, 3.0, 3.0, 0]}))
    zmax, passed = 0.0, False
    for _ in range(90):   # flight + the decaying bounces
        s.step(np.zeros((2, 8)))
        st = s.get_state()
        zmax = max(zmax, st[2])
        passed |= st[0] > 0.8
    assert zmax > 0.15 + 0.0215 and passed                  # cleared the 0.15 m robot
    assert np.all(s.get_state()[5 + 11 + 3: 5 + 11 + 5] == 0)   # the defender was never touched
    assert abs(s.get_state()[2] - 0.0215) < 1e-6            # back on the ground
    # the same kick flat (kick_v_z = 0) is blocked by the defender
    s = _ssl(backend, [0.1, 0.0, 0, 0], [0, 0, 0.0], ny=1, yellow=y)
    s.step(_cmd(2, 8, {0: [0, 0, 0, 0, 0, 3.0, 0.0, 0]}))
    for _ in range(40):
        s.step(np.zeros((2, 8)))
    assert s.get_state()[0] < 0.6


@pytest.mark.parametrize("backend", BACKENDS)
def test_ssl_ball_can_leave_the_field_lines(backend, oracle_mod):
    """static_defenders.py:185-197 tests |y| > W/2 and x > L/2 outside the goal: the SSL field
    has no wall on its lines, only further out."""
    f = _ssl(backend, [0, 0, 0, 0], [0, -1.5, 0]).get_field_params()
    s = _ssl(backend, [1.0, 1.9, 0.0, 2.0], [0, -1.5, 0])
    for _ in range(20):
        s.step(np.zeros((1, 8)))
    assert s.get_state()[1] > f["width"] / 2
    s = _ssl(backend, [2.8, 1.0, 2.0, 0.0], [0, -1.5, 0])
    for _ in range(20):
        s.step(np.zeros((1, 8)))
    assert s.get_state()[0] > f["length"] / 2


@pytest.mark.parametrize("backend", BACKENDS)
def test_time_step_scales_motion(backend, oracle_mod):
    a = _make(backend, 0, 0, 3, 3, 25)
    b = _make(backend, 0, 0, 3, 3, 50)
    for s in (a, b):
        s.reset(np.array([0.0, 0.5, 0.4, 0.0]), np.array([[-0.5, 0, 0], [-0.6, 0.5, 0], [-0.6, -0.5, 0]], float),
                np.array([[0.6, 0.5, 0], [0.6, 0, 0], [0.6, -0.5, 0]], float))
    for _ in range(4):
        a.step(np.zeros((6, 2)))
    for _ in range(2):
        b.step(np.zeros((6, 2)))
    # rolling resistance is applied once per step(): the two discretisations agree to O(mu dt T)
    assert np.allclose(a.get_state(), b.get_state(), atol=1e-3)


# ---- tangential friction, ball spin, second contact sweep ----
def _full(s):
    return s._sim.get_state_full()[0] if hasattr(s, "_sim") else s.o.get_state_full()


def _set_full(s, v):
    if hasattr(s, "_sim"):
        s._sim.set_state(np.asarray(v, float)[None])
    else:
        s.o.set_state_full(np.asarray(v, float))


@pytest.mark.parametrize("backend", BACKENDS)
def test_turning_robot_drags_the_ball_sideways_and_spins_it(backend, oracle_mod):
    """Coulomb friction at the contact point: a robot that hits the ball while turning drags it
    along its surface and spins it; the mirrored command gives the mirrored result."""
    out = {}
    for name, (wl, wr) in {"ccw": (12.0, 40.0), "cw": (40.0, 12.0), "straight": (30.0, 30.0)}.items():
        s = _vss(backend, [0.085, 0.0, 0, 0], [0.0, 0.0, 0.0])
        for _ in range(8):
            s.step(_cmd(6, 2, {0: [wl, wr]}))
        f = _full(s)
        out[name] = (f[1], f[4], f[-1])       # ball y, ball vy, ball spin
        assert f[3] > 0.05                     # it was hit
    y, vy, sp = out["ccw"]
    assert y > 1e-4 and vy > 1e-3              # counter-clockwise robot: its front surface moves to +y
    assert sp < -0.5                           # the ball spins the other way (gears)
    ym, vym, spm = out["cw"]
    assert abs(y + ym) < 1e-6 and abs(vy + vym) < 1e-5 and abs(sp + spm) < 1e-3
    assert abs(out["straight"][0]) < 1e-7 and abs(out["straight"][2]) < 1e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_ball_spin_couples_with_walls_and_decays(backend, oracle_mod):
    f = _vss(backend, [0, 0, 0, 0], [-0.5, 0.3, 0.0]).get_field_params()
    W = f["width"]
    # oblique bounce off the +y touch line: friction takes tangential speed and turns it into spin
    s = _vss(backend, [0.0, W / 2 - 0.06, 1.0, 1.0], [-0.5, -0.3, 0.0])
    for _ in range(6):
        s.step(_cmd(6, 2, {}))
    st = _full(s)
    assert st[4] < 0                                        # came back
    assert 0.5 < st[3] < 0.9                                # lost tangential speed beyond rolling resistance (0.045 m/s)
    assert st[-1] > 5.0                                     # and rolls along the wall: spin = counter-clockwise
    ke_in = 1.0 ** 2 + 1.0 ** 2
    assert st[3] ** 2 + st[4] ** 2 + 0.4 * (0.0215 * st[-1]) ** 2 < ke_in   # no energy from nowhere
    # a spinning ball hitting the wall head-on is deflected along it
    s = _vss(backend, [0.0, W / 2 - 0.06, 0.0, 1.0], [-0.5, -0.3, 0.0])
    v = _full(s); v[-1] = 60.0; _set_full(s, v)
    for _ in range(6):
        s.step(_cmd(6, 2, {}))
    st = _full(s)
    assert st[4] < 0 and st[3] > 0.05 and st[-1] < 60.0
    v = _full(s); v[-1] = -60.0; v[0:5] = [0.0, W / 2 - 0.06, 0.0215, 0.0, 1.0]; _set_full(s, v)
    for _ in range(6):
        s.step(_cmd(6, 2, {}))
    assert _full(s)[3] < -0.05
    # free spin decays at a constant rate to an exact stop
    s = _vss(backend, [0, 0, 0, 0], [-0.5, 0.3, 0.0])
    v = _full(s); v[-1] = 20.0; _set_full(s, v)
    prev = 20.0
    for t in range(40):
        s.step(_cmd(6, 2, {}))
        sp = _full(s)[-1]
        assert 0.0 <= sp < prev or sp == prev == 0.0
        prev = sp
    assert prev == 0.0 and abs(_full(s)[0]) < 1e-9         # spin alone does not move the ball


@pytest.mark.parametrize("backend", BACKENDS)
def test_oblique_hit_on_a_resting_robot_loses_tangential_speed(backend, oracle_mod):
    """ball thrown past the edge of a parked SSL robot: the grazing contact reduces the speed
    along the surface and leaves the ball spinning; the reflected normal speed obeys e_rb."""
    s = _ssl(backend, [-0.5, 0.095, 2.0, 0.0], [0.0, 0.0, 180.0])   # kicker looks away: body-circle contact
    for _ in range(14):
        s.step(_cmd(1, 8, {}))
    st = _full(s)
    speed = math.hypot(st[3], st[4])
    assert st[4] > 0.1                       # deflected to +y
    assert speed < 1.7                       # 2.0 - rolling resistance (14 * 0.01) would be 1.86 without the contact
    assert abs(st[-1]) > 1.0                 # spinning


@pytest.mark.parametrize("backend", BACKENDS)
def test_crowded_pushing_keeps_penetration_small(backend, oracle_mod):
    """22 SSL robots all driving at the field centre at 2 m/s (two Jacobi sweeps per sub-step when
    anything touches): overlap of any pair stays below 5 mm (2.8 % of a diameter) after every step;
    a ball squeezed against a wall by a VSS robot stays within 3 mm of touching."""
    s = _make(backend, 1, 1, 11, 11)
    rng = np.random.default_rng(0)
    pts = [(0.19 * (i - 2.5) + rng.uniform(-.004, .004), 0.19 * (j - 2) + rng.uniform(-.004, .004))
           for i in range(6) for j in range(4)][:22]
    pose = np.array([[p[0], p[1], rng.uniform(-180, 180)] for p in pts])
    s.reset(np.array([2.0, 2.0, 0, 0.0]), pose[:11], pose[11:])
    worst = 0.0
    for _ in range(60):
        st = s.get_state()
        cmds = np.zeros((22, 8))
        for k in range(22):
            x, y, th = st[5 + 11 * k], st[6 + 11 * k], math.radians(st[7 + 11 * k])
            n = math.hypot(x, y) + 1e-9
            gx, gy = -2.0 * x / n, -2.0 * y / n
            cmds[k, 1] = gx * math.cos(th) + gy * math.sin(th)
            cmds[k, 2] = -gx * math.sin(th) + gy * math.cos(th)
        s.step(cmds)
        st = s.get_state()
        P = st[5:].reshape(22, 11)[:, :2]
        D = np.hypot(*(P[:, None] - P[None]).transpose(2, 0, 1))
        np.fill_diagonal(D, 9.0)
        worst = max(worst, 0.18 - D.min())
    assert 0.0 < worst < 0.005, worst
    # VSS: robot pins the ball against the +x goal-line wall (outside the goal mouth)
    f = _vss(backend, [0, 0, 0, 0], [-0.5, 0.3, 0.0]).get_field_params()
    L = f["length"]
    s = _vss(backend, [L / 2 - 0.03, 0.4, 0, 0], [L / 2 - 0.12, 0.4, 0.0])
    for _ in range(40):
        s.step(_cmd(6, 2, {0: [40.0, 40.0]}))
        st = s.get_state()
        assert math.hypot(st[0] - st[5], st[1] - st[6]) > 0.0375 + 0.0215 - 3e-3
        assert st[0] <= L / 2 - 0.0215 + 1e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_contacts_never_add_kinetic_energy(backend, oracle_mod):
    """robots thrown at each other with the motors commanded to stop: every mechanism on the path
    (restitution < 1, Coulomb friction, braking, positional de-penetration) is dissipative, so the
    kinetic energy of the env never rises from one step to the next."""
    s = _make(backend, 1, 1, 11, 11)
    rng = np.random.default_rng(4)
    pts = [(0.21 * (i - 2.5), 0.21 * (j - 2)) for i in range(6) for j in range(4)][:22]
    pose = np.array([[p[0], p[1], rng.uniform(-180, 180)] for p in pts])
    s.reset(np.array([0.0, 3.0, 0.5, -3.0]), pose[:11], pose[11:])
    v = _full(s)
    for k in range(22):
        v[5 + 11 * k + 3: 5 + 11 * k + 5] = rng.uniform(-2.0, 2.0, 2)
    _set_full(s, v)

    def ke(st):
        R = st[5:5 + 22 * 11].reshape(22, 11)
        return 0.5 * 2.2 * (R[:, 3:5] ** 2).sum() + 0.5 * 0.046 * (st[3] ** 2 + st[4] ** 2 + 0.4 * (0.0215 * st[-1]) ** 2)
    prev = ke(_full(s))
    touched = False
    for _ in range(50):
        s.step(np.zeros((22, 8)))
        st = _full(s)
        P = st[5:5 + 22 * 11].reshape(22, 11)[:, :2]
        D = np.hypot(*(P[:, None] - P[None]).transpose(2, 0, 1)); np.fill_diagonal(D, 9.0)
        touched |= D.min() < 0.1805
        cur = ke(st)
        assert cur <= prev * (1 + 1e-6) + 1e-9, (cur, prev)
        prev = cur
    assert touched and prev < 0.02          # only the ball still rolls


# ---- model v2: wall-aware robot-robot contacts, goal posts (SSL) ----

def _ssl_state(st, n):
    return st[0:2], np.stack([st[5 + 11 * k: 7 + 11 * k] for k in range(n)])


@pytest.mark.parametrize("backend", BACKENDS)
def test_goal_post_is_a_point_bodies_keep_their_radius_from(backend, oracle_mod):
    """The open end of a goal's side wall at (L/2, goal_width/2).  A ball shot at it comes back (radial speed reversed, scaled by the
    wall restitution) and never gets closer than its radius; a robot driven along the goal line through the post's neighbourhood slides
    AROUND it: never closer than its radius, and never displaced by more than it can drive in a step (v1 threw a robot that had
    overlapped the wall's end 4 cm sideways when it crossed the goal line)."""
    s = _ssl(backend, [0.0, 0.0, 0.0, 0.0], [0.0, 0.0, 0.0], ft=2)
    f = s.get_field_params()
    hl, ghw, rb, rr = f["length"] / 2, f["goal_width"] / 2, f["ball_radius"], f["rbt_radius"]
    post = np.array([hl, ghw])
    # ball: from the field side, straight at the post, 45 degrees
    start = post - np.array([0.3, 0.3])
    s.reset(np.array([start[0], start[1], 2.0, 2.0]), np.array([[-1.0, -1.0, 0.0]]), np.zeros((0, 3)))
    dmin, back = 9.0, False
    for _ in range(20):
        s.step(np.zeros((1, 8)))
        st = s.get_state()
        dmin = min(dmin, float(np.hypot(*(st[0:2] - post))))
        back = back or (st[3] < 0.0 and st[4] < 0.0)
    assert dmin >= rb - 1e-6 and back
    # robot: along the goal line (x = L/2 - 0.03) towards the goal mouth, through the post's neighbourhood
    s.reset(np.array([0.0, 0.0, 0.0, 0.0]), np.array([[hl - 0.03, ghw + 0.4, 0.0]]), np.zeros((0, 3)))
    prev, dmin = None, 9.0
    for _ in range(60):
        s.step(_cmd(1, 8, {0: [0, 0.0, -1.5, 0.0]}))      # robot-local v_y = -1.5 m/s (heading 0: global -y)
        st = s.get_state()
        p = st[5:7].copy()
        dmin = min(dmin, float(np.hypot(*(p - post))))
        if prev is not None:
            assert np.hypot(*(p - prev)) <= 1.6 * 0.025 + 1e-3, (prev, p)
        prev = p
    assert dmin >= rr - 1e-5
    assert prev[1] < ghw - rr + 0.02          # it got past the post into the mouth's height


@pytest.mark.parametrize("backend", BACKENDS)
def test_a_row_of_pushers_against_a_wall_does_not_telescope(backend, oracle_mod):
    """Six robots in a row perpendicular to a side wall, the first one standing at the wall, all driving at it at full speed for a second
    and a half: the wall holds the first one, so its neighbour takes the whole correction of their contact (and so on down the row, a
    third and fourth sweep while the pile is deep) — no pair overlaps by more than 8 mm and nobody is beyond the wall.  (Model v1, which
    split every correction in halves and clamped the outer robot back into its neighbour afterwards: 13 mm here; both: 5 mm in the open.)"""
    n = 6
    s = _make(backend, 1, 2, n, 0)
    f = s.get_field_params()
    hw, r = f["width"] / 2, f["rbt_radius"]
    yl = hw + 0.3 - r
    s.reset(np.array([-2.0, 0.0, 0.0, 0.0]), np.array([[0.5, yl - 0.001 - k * (2 * r + 0.005), 0.0] for k in range(n)]), np.zeros((0, 3)))
    worst = 0.0
    for t in range(60):
        s.step(_cmd(n, 8, {k: [0, 0.0, 2.5, 0.0] for k in range(n)}))      # robot-local v_y (heading 0: global +y, at the wall)
        _, p = _ssl_state(s.get_state(), n)
        assert (p[:, 1] <= yl + 1e-5).all()
        if t >= 10:
            d = np.hypot(p[:, None, 0] - p[None, :, 0], p[:, None, 1] - p[None, :, 1]) + 9.0 * np.eye(n)
            worst = max(worst, 2 * r - float(d.min()))
    assert 0.002 < worst < 0.008, worst


@pytest.mark.parametrize("backend", BACKENDS)
def test_a_pile_driven_into_a_corner_stays_a_pile_of_discs(backend, oracle_mod):
    """Six robots driving into a field corner for two seconds (both wall normals blocked for the first one, a chain of pushers
    behind it): no pair overlaps by more than 1.5 mm at any step after the pile has formed (model v1: 2 mm)."""
    s = _make(backend, 1, 2, 6, 0)
    f = s.get_field_params()
    hl, hw, r = f["length"] / 2, f["width"] / 2, f["rbt_radius"]
    xl, yl = hl + 0.3 - r, hw + 0.3 - r
    pos = [[xl - 0.05 - 0.22 * (k % 3), yl - 0.05 - 0.22 * (k // 3), 30.0 * k] for k in range(6)]
    s.reset(np.array([-2.0, 0.0, 0.0, 0.0]), np.array(pos), np.zeros((0, 3)))
    worst = 0.0
    for t in range(80):
        st = s.get_state()
        th = np.deg2rad(np.array([st[7 + 11 * k] for k in range(6)]))
        cm = np.zeros((6, 8))
        cm[:, 1] = 1.5 * (np.cos(th) + np.sin(th)); cm[:, 2] = 1.5 * (-np.sin(th) + np.cos(th))    # global (+1.5, +1.5) in the robot frame
        s.step(cm)
        st = s.get_state()
        _, p = _ssl_state(st, 6)
        assert (np.abs(p[:, 0]) <= xl + 1e-5).all() and (np.abs(p[:, 1]) <= yl + 1e-5).all()
        if t >= 20:
            d = np.hypot(p[:, None, 0] - p[None, :, 0], p[:, None, 1] - p[None, :, 1]) + 9.0 * np.eye(6)
            worst = max(worst, 2 * r - float(d.min()))
    assert 0.0 < worst < 0.0015, worst


# ---- model v2 of the VSS class (round 6): goal posts as chords, held axes ----
def _vss_state(st, n):
    return np.array([[st[5 + 6 * k], st[6 + 6 * k]] for k in range(n)])


@pytest.mark.parametrize("backend", BACKENDS)
def test_vss_goal_post_is_a_chord_bodies_slide_along(backend, oracle_mod):
    """The corner where the goal line's wall ends at the goal mouth, (L/2, goal_width/2).  In the corner region (|x| <= L/2,
    |y| < goal_width/2) a body keeps |x| + |y| <= L/2 + goal_width/2 - r: the chord of the arc a round post would give.  A ball shot
    diagonally at the corner comes back (the velocity component along the chord's normal is reflected with the wall restitution); a
    robot that slides down the goal line's wall towards the mouth goes round the corner without ever being thrown: its displacement
    per step stays what it can drive (model v1 clamped x to the goal line's limit for |y| > goal_width/2 - r and threw a body that
    entered that strip from inside the mouth up to a radius — 3.75 cm — sideways)."""
    s = _vss(backend, [0, 0, 0, 0], [0, 0.5, 0])
    f = s.get_field_params()
    hl, ghw, rb, rr = f["length"] / 2, f["goal_width"] / 2, f["ball_radius"], f["rbt_radius"]
    # ball: from the field side, 45 degrees, aimed at the chord's middle
    start = np.array([hl, ghw]) - np.array([0.25, 0.25]) - np.array([rb, rb]) / 2
    s.reset(np.array([start[0], start[1], 1.5, 1.5]), np.array([[-0.5, -0.5, 0.0], [-0.6, 0.5, 0.0], [-0.6, -0.3, 0.0]]),
            np.array([[-0.2, 0.5, 0.0], [-0.2, 0.0, 0.0], [-0.2, -0.5, 0.0]]))
    back, worst = False, -9.0
    for _ in range(16):
        s.step(np.zeros((6, 2)))
        st = s.get_state()
        if abs(st[0]) <= hl and abs(st[1]) < ghw:
            worst = max(worst, abs(st[0]) + abs(st[1]) - (hl + ghw - rb))
        back = back or (st[3] < 0.0 and st[4] < 0.0)
    assert back and worst <= 1e-6, (back, worst)
    assert abs(st[3] - st[4]) < 0.05 and -1.5 * 0.6 - 0.1 < st[3] < -0.5      # came back along the diagonal, slower (restitution 0.6, rolling resistance)
    # robot: pressed against the goal line's wall above the mouth, driving down along it (heading -80 degrees: a little into the wall)
    s.reset(np.array([-0.5, 0.0, 0.0, 0.0]), np.array([[hl - rr, ghw + 0.12, -80.0], [-0.6, 0.5, 0.0], [-0.6, -0.3, 0.0]]),
            np.array([[-0.2, 0.5, 0.0], [-0.2, 0.0, 0.0], [-0.2, -0.5, 0.0]]))
    prev, worst, entered = None, -9.0, False
    w = 0.5 / 0.026                                                            # wheel speed of 0.5 m/s
    for _ in range(40):
        s.step(_cmd(6, 2, {0: [w, w]}))
        st = s.get_state()
        p = st[5:7].copy()
        if abs(p[0]) <= hl and abs(p[1]) < ghw:
            worst = max(worst, abs(p[0]) + abs(p[1]) - (hl + ghw - rr))
        entered = entered or p[0] > hl - rr + 0.01                             # beyond the goal line's limit: inside the mouth
        if prev is not None:
            assert np.hypot(*(p - prev)) <= 0.5 * 0.025 * 1.3 + 5e-4, (prev, p)
        prev = p
    assert worst <= 1e-6 and entered, (worst, entered, prev)


@pytest.mark.parametrize("backend", BACKENDS)
def test_vss_row_of_pushers_against_a_wall_does_not_telescope(backend, oracle_mod):
    """Three VSS robots in a row perpendicular to a touch line, the first one standing at it, all driving at it at full speed for a
    second and a half: the wall holds the first one on the y axis (held axes: its centre is within 1 mm of the wall clamp's limit), so
    its neighbour takes the whole correction of their contact, and so on down the row — no pair overlaps by more than 3 mm and nobody
    is beyond the wall.  (Model v1, which split every correction in halves and clamped the outer robot back into its neighbour
    afterwards: 4.5 mm; model v2: 1.9 mm.)"""
    s = _make(backend, 0, 0, 3, 3)
    f = s.get_field_params()
    hw, r = f["width"] / 2, f["rbt_radius"]
    yl = hw - r
    blue = [[0.1, yl - 0.0005 - k * (2 * r + 0.003), 90.0] for k in range(3)]
    s.reset(np.array([-0.5, -0.4, 0.0, 0.0]), np.array(blue), np.array([[-0.5, 0.3, 0.0], [-0.5, 0.0, 0.0], [-0.3, -0.3, 0.0]]))
    worst = 0.0
    for t in range(60):
        s.step(_cmd(6, 2, {k: [46.0, 46.0] for k in range(3)}))
        p = _vss_state(s.get_state(), 3)
        assert (p[:, 1] <= yl + 1e-6).all()
        if t >= 10:
            d = np.hypot(p[:, None, 0] - p[None, :, 0], p[:, None, 1] - p[None, :, 1]) + 9.0 * np.eye(3)
            worst = max(worst, 2 * r - float(d.min()))
    assert 0.0005 < worst < 0.003, worst


@pytest.mark.parametrize("backend", BACKENDS)
def test_vss_pile_in_a_corner_and_in_a_goal_box_stays_a_pile_of_discs(backend, oracle_mod):
    """Three robots driving into a field corner, three into a goal box (7.5 cm robots in a 10 cm deep, 40 cm wide box: back wall and side
    walls hold on both axes), for two seconds: inside the walls, and no pair overlaps by more than 3 mm once the piles have formed."""
    s = _make(backend, 0, 0, 3, 3)
    f = s.get_field_params()
    hl, hw, ghw, gd, r = f["length"] / 2, f["width"] / 2, f["goal_width"] / 2, f["goal_depth"], f["rbt_radius"]
    blue = [[-(hl - 0.10 - 0.09 * k), hw - 0.08 - 0.05 * k, 135.0] for k in range(3)]                  # at the corner (-L/2, +W/2)
    yel = [[hl - 0.12, 0.02, 0.0], [hl - 0.21, -0.05, 0.0], [hl - 0.21, 0.09, 0.0]]                   # in front of the goal at +L/2
    s.reset(np.array([0.0, -0.5, 0.0, 0.0]), np.array(blue), np.array(yel))
    worst = 0.0
    for t in range(80):
        st = s.get_state()
        cm = np.zeros((6, 2))
        for k in range(6):
            x, y, th = st[5 + 6 * k], st[6 + 6 * k], np.deg2rad(st[7 + 6 * k])
            tx, ty = (-hl - 0.5, hw + 0.5) if k < 3 else (hl + gd + 0.5, 0.0)
            err = np.arctan2(ty - y, tx - x) - th
            err = np.arctan2(np.sin(err), np.cos(err))
            v, w = 0.8, 8.0 * err
            cm[k] = [(v - w * 0.04) / 0.026, (v + w * 0.04) / 0.026]
        s.step(cm)
        p = _vss_state(s.get_state(), 6)
        assert (np.abs(p[:, 0]) <= hl + gd - r + 1e-6).all() and (np.abs(p[:, 1]) <= hw - r + 1e-6).all()
        inside = np.abs(p[:, 0]) > hl
        assert (np.abs(p[inside, 1]) <= ghw - r + 1e-6).all()
        if t >= 25:
            d = np.hypot(p[:, None, 0] - p[None, :, 0], p[:, None, 1] - p[None, :, 1]) + 9.0 * np.eye(6)
            worst = max(worst, 2 * r - float(d.min()))
    assert 0.0 < worst < 0.003, worst
