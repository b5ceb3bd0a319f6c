"""The array-form pieces of the Python host layer against the scalar forms they replace (CPU only): the OU bank against
one process at a time (Utils/Utils.py:14-21 of the reference, same global numpy stream), wire-format commands against the
List[Robot] packing of rsim.py:91-102, the lazily built Frame records against an eager parse (Entities/Frame.py:18-49),
and observation plans against the per-value hooks (vss_gym.py:93-117, static_defenders.py:90-112)."""
import numpy as np
import pytest

import fake_robosim


def test_ou_bank_equals_sampling_one_process_at_a_time(oracle_mod):
    from rsoccer_amd import gymshim as gym
    from rsoccer_amd.Utils import OrnsteinUhlenbeckAction, OrnsteinUhlenbeckBank
    box = gym.spaces.Box(low=-1, high=1, shape=(2,), dtype=np.float32)
    np.random.seed(7)
    solo = [OrnsteinUhlenbeckAction(box, dt=0.025) for _ in range(5)]
    want = np.array([[p.sample() for p in solo] for _ in range(40)])
    np.random.seed(7)
    procs = [OrnsteinUhlenbeckAction(box, dt=0.025) for _ in range(5)]
    bank = OrnsteinUhlenbeckBank(procs)
    got = np.array([bank.sample() for _ in range(40)])
    assert got.dtype == np.float64 and np.array_equal(got, want)
    assert all(np.array_equal(p.x_prev, got[-1][i]) for i, p in enumerate(procs))   # the processes own their state
    # a reset, or a sample taken by hand, in between is honoured by the next bank step
    np.random.seed(11)
    for p in solo:
        p.reset()
    solo[2].sample()
    want2 = np.array([p.sample() for p in solo])
    np.random.seed(11)
    for p in procs:
        p.reset()
    procs[2].sample()
    assert np.array_equal(bank.sample(), want2)
    with pytest.raises(ValueError):
        OrnsteinUhlenbeckBank([OrnsteinUhlenbeckAction(box, dt=0.025), OrnsteinUhlenbeckAction(box, dt=0.05)])


def test_command_rows_are_the_robot_list_in_wire_format(oracle_mod):
    from rsoccer_amd.Entities import Robot
    from rsoccer_amd.Simulators.rsim import CommandRows
    from rsoccer_amd.vss.env_vss import VSSEnv
    fake_robosim.arm()
    env = VSSEnv(sim_backend=fake_robosim)
    env.reset()
    np.random.seed(3)
    action = np.array([0.3, -0.02], dtype=np.float32)
    cmds = env._get_commands(action)
    assert type(cmds) is CommandRows and len(cmds) == 6 and cmds.rows.shape == (6, 2) and cmds.rows.dtype == np.float64
    robots = list(cmds)
    assert all(isinstance(r, Robot) for r in robots) and cmds[0] is robots[0]
    assert [(r.yellow, r.id) for r in robots] == [(False, 0), (False, 1), (False, 2), (True, 0), (True, 1), (True, 2)]
    assert robots[0].v_wheel0.dtype == np.float32 and robots[0].v_wheel1 == 0       # the agent keeps the action's dtype; dead zone
    sent = []
    env.rsim.simulator.step = lambda c: sent.append(np.array(c))
    env.rsim.send_commands(cmds)                       # wire format goes through untouched ...
    env.rsim.send_commands(robots)                     # ... and equals the packing of the same records
    assert np.array_equal(sent[0], sent[1]) and sent[0] is not cmds.rows
    # the same numbers as the per-robot scalar form
    left, right = env._actions_to_v_wheels(action)
    assert cmds.rows[0, 0] == np.float64(left) and cmds.rows[0, 1] == np.float64(right)
    env.close()


def test_frame_records_are_built_on_first_use(oracle_mod):
    from rsoccer_amd.Entities import Ball, Frame, FrameSSL, Robot
    state = np.arange(5 + 11 * 3, dtype=np.float64) * 0.5
    state[5 + 6] = 0.0                                   # blue 0: infrared off
    f = FrameSSL().parse(state, 1, 2)
    assert f.state is state and "ball" not in f.__dict__ and "robots_blue" not in f.__dict__
    assert f.robots_yellow[1].v_wheel3 == state[5 + 22 + 10] and "ball" in f.__dict__
    assert f.robots_blue[0].infrared is False and f.robots_yellow[0].infrared is True and f.ball.v_y == state[4]
    f.parse(state * 2, 1, 2)                             # parsing again drops the old records
    assert "ball" not in f.__dict__ and f.ball.x == 0.0 and f.robots_yellow[0].x == state[16] * 2
    hand = Frame()                                       # a frame assembled by hand keeps what it is given
    hand.ball = Ball(x=1.0, y=2.0)
    hand.robots_blue[0] = Robot(x=3.0)
    assert hand.state is None and hand.ball.x == 1.0 and hand.robots_yellow == {} and hand.robots_blue[0].x == 3.0
    with pytest.raises(AttributeError):
        hand.no_such_field
    with pytest.raises(NotImplementedError):
        Frame().parse(state)


@pytest.mark.parametrize("which", ["vss", "sd", "drib", "cont", "pass"])
def test_observation_plans_equal_the_scalar_hooks(oracle_mod, which):
    """the array-form observation against the per-value path the same classes fall back to for a hand-made frame"""
    from rsoccer_amd.Entities import Ball, Frame, Robot
    from rsoccer_amd.ssl.ssl_hw_challenge import (SSLContestedPossessionEnv, SSLHWDribblingEnv, SSLHWStaticDefendersEnv,
                                                  SSLPassEnduranceEnv)
    from rsoccer_amd.vss.env_vss import VSSEnv
    fake_robosim.arm()
    env = {"vss": lambda: VSSEnv(sim_backend=fake_robosim), "sd": lambda: SSLHWStaticDefendersEnv(sim_backend=fake_robosim),
           "drib": lambda: SSLHWDribblingEnv(sim_backend=fake_robosim), "cont": lambda: SSLContestedPossessionEnv(sim_backend=fake_robosim),
           "pass": lambda: SSLPassEnduranceEnv(sim_backend=fake_robosim)}[which]()
    rng = np.random.default_rng(5)
    env.reset()
    wide = 6 if which == "vss" else 11
    n = env.n_robots_blue + env.n_robots_yellow
    for _ in range(20):
        state = rng.uniform(-4, 4, 5 + wide * n)
        state[7::wide] = rng.uniform(-400, 400, n)               # headings in degrees, beyond one turn too
        state[10::wide] = rng.uniform(-900, 900, n)              # deg/s: saturates the normaliser
        if wide == 11:
            state[11::wide] = rng.integers(0, 2, n)              # infrared
        env.frame = type(env.rsim.get_frame())().parse(state, env.n_robots_blue, env.n_robots_yellow)
        fast = env._frame_to_observations()
        hand = Frame()                                           # the same world as records only: the scalar path
        hand.ball = Ball(x=state[0], y=state[1], z=state[2], v_x=state[3], v_y=state[4])
        for k in range(n):
            b = 5 + wide * k
            r = Robot(x=state[b], y=state[b + 1], theta=state[b + 2], v_x=state[b + 3], v_y=state[b + 4], v_theta=state[b + 5])
            if wide == 11:
                r.infrared = bool(state[b + 6])
            (hand.robots_blue if k < env.n_robots_blue else hand.robots_yellow)[k if k < env.n_robots_blue else k - env.n_robots_blue] = r
        env.frame = hand
        slow = env._frame_to_observations()
        assert fast.dtype == np.float32 and fast.shape == env.observation_space.shape and np.array_equal(fast, slow), which
    env.close()
