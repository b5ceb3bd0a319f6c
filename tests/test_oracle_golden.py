"""Pins the CPU oracle's TASK ARITHMETIC to the reference.

Golden vectors: tests/golden/reference_vectors.npz, produced by tests/golden/make_golden.py by
importing the reference's own VSSEnv / SSLHWStaticDefendersEnv / Frame / RSim / OU / KDTree code
(/root/reference, build container only).  The oracle's physics has no reference counterpart to
be pinned against (rc-robosim is absent) — see oracle/rsx_oracle.c; for it this file only holds
a regression check against trajectories recorded from the oracle itself.
"""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
PRECS = [("f64", 2e-7, 1e-9), ("f32", 3e-6, 2e-5)]  # (precision, obs tolerance, scalar tolerance)


def _env(O, task, prec):
    e = O.OracleEnv(0, 0, 3, 3, 25, prec) if task == 1 else O.OracleEnv(1, 2, 1, 6, 25, prec)
    e.task_attach(task, 0, 0, 0)
    return e


def test_field_tables_match_fixture(oracle_mod):
    assert np.array_equal(oracle_mod.OracleEnv(0, 0, 3, 3).field_params(), G["vss_field"])
    assert np.array_equal(oracle_mod.OracleEnv(1, 2, 1, 6).field_params(), G["ssl_field"])


@pytest.mark.parametrize("prec,otol,stol", PRECS)
def test_normalisers(oracle_mod, prec, otol, stol):
    # vss_gym_base.py:52-58, static_defenders.py:76-77
    assert np.allclose(_env(oracle_mod, 1, prec).norms(), G["vss_norms"], rtol=1e-6 if prec == "f32" else 1e-12)
    assert np.allclose(_env(oracle_mod, 2, prec).norms(), G["sd_norms"], rtol=1e-6 if prec == "f32" else 1e-12)


@pytest.mark.parametrize("prec,otol,stol", PRECS)
@pytest.mark.parametrize("task,key", [(1, "vss"), (2, "sd")])
def test_observations(oracle_mod, prec, otol, stol, task, key):
    e = _env(oracle_mod, task, prec)
    for s, want in zip(G[f"{key}_obs_states"], G[f"{key}_obs"]):
        e.set_state_full(np.append(s, [0.0, 0.0]))
        got = e.obs_eval()
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) <= otol


@pytest.mark.parametrize("prec,otol,stol", PRECS)
def test_vss_wheel_commands(oracle_mod, prec, otol, stol):
    e = _env(oracle_mod, 1, prec)
    for a, want in zip(G["vss_wheel_actions"], G["vss_wheel_cmds"]):
        act = np.zeros((6, 2)); act[0] = a
        got = e.cmds_eval(act)[0]
        assert np.allclose(got, want, rtol=stol, atol=stol), (a, got, want)
    # the dead zone really zeroes (vss_gym.py:244-248)
    act = np.zeros((6, 2)); act[0] = [0.04, -0.04]
    assert np.array_equal(e.cmds_eval(act)[0], [0.0, 0.0])


@pytest.mark.parametrize("prec,otol,stol", PRECS)
def test_static_defenders_commands(oracle_mod, prec, otol, stol):
    e = _env(oracle_mod, 2, prec)
    for a, th, want in zip(G["sd_cmd_actions"], G["sd_cmd_thetas"], G["sd_cmds"]):
        got = e.cmds_eval(a, th)
        assert np.allclose(got, want, rtol=0, atol=max(stol, 1e-9) * 10), (a, th, got[0], want[0])
        assert np.all(got[1:] == 0)


@pytest.mark.parametrize("prec,otol,stol", PRECS)
def test_ou_process(oracle_mod, prec, otol, stol):
    e = _env(oracle_mod, 1, prec)
    x = np.zeros(2)
    for n, want in zip(G["ou_normals"], G["ou_samples"]):
        x = e.ou_eval(x, n)
        assert np.allclose(x, want, rtol=0, atol=2e-6)  # the reference itself mixes f32 and f64 here


def _episodes(prefix):
    n = int(G[f"{prefix}_n_episodes"])
    for i in range(n):
        yield i, {k: G[f"{prefix}_ep{i}_{k}"] for k in ("reset_state", "obs0", "actions", "cmds", "states", "obs", "reward", "done", "info")}


@pytest.mark.parametrize("prec,otol,stol", PRECS)
@pytest.mark.parametrize("task,prefix", [(1, "vss"), (2, "sd")])
def test_reward_done_info_over_episodes(oracle_mod, prec, otol, stol, task, prefix):
    """obs / reward / done / cumulative info of every recorded transition (goals, rule
    infractions and first-step behaviour included)."""
    seen_done = 0
    for i, ep in _episodes(prefix):
        e = _env(oracle_mod, task, prec)
        e.set_state_full(np.append(ep["reset_state"], [0.0, 0.0]))
        assert np.max(np.abs(e.obs_eval() - ep["obs0"])) <= otol
        last = ep["reset_state"]
        for t in range(len(ep["reward"])):
            e.set_state_full(np.append(ep["states"][t], [0.0, 0.0]))
            assert np.max(np.abs(e.obs_eval() - ep["obs"][t])) <= otol, (i, t)
            r, d = e.reward_eval(last, ep["cmds"][t], t == 0)
            assert d == bool(ep["done"][t]), (i, t)
            # float32 actions make the reference evaluate its energy term in float32 (numpy promotion)
            assert abs(r - ep["reward"][t]) <= (5e-5 if prec == "f32" else 2e-8), (i, t, r, ep["reward"][t])
            info = e.task_out()["info"]
            assert np.allclose(info, ep["info"][t], rtol=0, atol=2e-4 if prec == "f32" else 1e-6), (i, t, info, ep["info"][t])
            last = ep["states"][t]
            seen_done += int(d)
    assert seen_done >= (2 if task == 1 else 6)  # the fixtures do exercise the terminal branches


@pytest.mark.parametrize("task,prefix", [(1, "vss"), (2, "sd")])
def test_agent_command_from_action_in_episodes(oracle_mod, task, prefix):
    e = _env(oracle_mod, task, "f64")
    for i, ep in _episodes(prefix):
        last = ep["reset_state"]
        for t in range(len(ep["reward"])):
            if task == 1:
                act = np.zeros((6, 2)); act[0] = ep["actions"][t]
                # float32 actions: the reference's wheel mapping then runs in float32
                assert np.allclose(e.cmds_eval(act)[0], ep["cmds"][t][0], rtol=3e-7, atol=1e-6)
            else:
                # float32 actions: the reference then multiplies in float32 (numpy weak-scalar promotion)
                assert np.allclose(e.cmds_eval(ep["actions"][t], last[7])[0], ep["cmds"][t][0], rtol=0, atol=2e-6)
            last = ep["states"][t]


@pytest.mark.parametrize("task,prefix,kind", [(1, "vss", (0, 0, 3, 3)), (2, "sd", (1, 2, 1, 6))])
def test_physics_regression_against_recorded_oracle_runs(oracle_mod, task, prefix, kind):
    """NOT a reference pin: the recorded state sequences came from this oracle (f64).  Guards the
    model against silent changes — a deliberate model change must regenerate the fixtures."""
    for i, ep in _episodes(prefix):
        e = oracle_mod.OracleEnv(*kind, 25, "f64")
        e.set_state_full(np.append(ep["reset_state"], [0.0, 0.0]))
        for t in range(len(ep["reward"]) - 1):  # the last state of scripted episodes is injected
            e.step(ep["cmds"][t])
            assert np.allclose(e.get_state(), ep["states"][t], rtol=0, atol=1e-12), (i, t)


def test_philox_known_answers(oracle_mod):
    # Random123 known-answer vectors for philox4x32-10
    P = oracle_mod.philox
    assert P([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert P([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert P([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # the engine draws with the 7-round member of the family (same round function, pinned above)
    assert P([0, 0, 0, 0], [0, 0], 7) == [0x5f6fb709, 0x0d893f64, 0x4f121f81, 0x4f730a48]


def test_elementary_functions(oracle_mod):
    import math
    for a in np.linspace(-3.3, 3.3, 4001):
        s, c = oracle_mod.sincos(a, "f32")
        a32 = float(np.float32(a))
        assert abs(s - math.sin(a32)) < 2.5e-7 and abs(c - math.cos(a32)) < 2.5e-7
    for x in np.linspace(2.0 ** -24, 1.0, 4001):
        x32 = float(np.float32(x))
        assert abs(oracle_mod.log(x, "f32") - math.log(x32)) < 4e-7 * max(1.0, abs(math.log(x32)))


@pytest.mark.parametrize("task,kind,min_d", [(1, (0, 0, 3, 3), 0.1), (2, (1, 2, 1, 6), 0.2)])
def test_device_style_placement_obeys_the_reference_rules(oracle_mod, task, kind, min_d):
    """vss_gym.py:194-233 / static_defenders.py:214-254 with Philox draws: ranges and spacing."""
    f = oracle_mod.OracleEnv(*kind).field_params()
    L, W, pen_l, pen_w = f[0], f[1], f[2], f[3]
    for env_id in range(200):
        e = oracle_mod.OracleEnv(*kind, 25, "f32")
        e.task_attach(task, 42, env_id, 0)
        e.task_reset()
        s = e.get_state()
        rs = 6 if task == 1 else 11
        pts = [s[0:2]] + [s[5 + rs * k: 7 + rs * k] for k in range(kind[2] + kind[3])]
        th = [s[5 + rs * k + 2] for k in range(kind[2] + kind[3])]
        for i in range(len(pts)):
            for j in range(i):
                assert np.hypot(*(pts[i] - pts[j])) >= min_d - 1e-6
        assert all(0 <= t < 360 for t in th)
        if task == 1:
            assert all(abs(p[0]) <= L / 2 - 0.1 + 1e-6 and abs(p[1]) <= W / 2 - 0.1 + 1e-6 for p in pts)
        else:
            assert np.allclose(pts[1], 0) and th[0] == 0
            for p in [pts[0]] + pts[2:]:
                assert 0.2 - 1e-6 <= p[0] <= L / 2 - 0.1 + 1e-6 and abs(p[1]) <= W / 2 - 0.1 + 1e-6
            assert not (pts[0][0] > L / 2 - pen_l and abs(pts[0][1]) < pen_w / 2)
        assert np.all(s[3:5] == 0)


# ---------------------------------------------------------------------------------------------
# SSLDribbling-v0 (task 3), SSLContestedPossession-v0 (task 4), SSLPassEndurance-v0 (task 5)
# ---------------------------------------------------------------------------------------------
OTHER = {"drib": (3, (1, 2, 1, 4)), "cont": (4, (1, 2, 1, 1)), "pass": (5, (1, 2, 2, 0))}


def _other_env(O, tag, prec):
    task, kind = OTHER[tag]
    e = O.OracleEnv(*kind, 25, prec)
    e.task_attach(task, 0, 0, 0)
    return e


@pytest.mark.parametrize("prec,otol,stol", PRECS)
@pytest.mark.parametrize("tag", sorted(OTHER))
def test_other_tasks_observations(oracle_mod, prec, otol, stol, tag):
    e = _other_env(oracle_mod, tag, prec)
    assert np.allclose(e.norms(), G[f"{tag}_norms"], rtol=1e-6)
    for i, (s, want) in enumerate(zip(G[f"{tag}_obs_states"], G[f"{tag}_obs"])):
        e.set_state_full(np.append(s, [0.0, 0.0]))
        if tag == "drib":
            e.set_scalar(i % 7)
        got = e.obs_eval()
        assert got.shape == want.shape and np.max(np.abs(got - want)) <= otol, (tag, i)


@pytest.mark.parametrize("prec,otol,stol", PRECS)
@pytest.mark.parametrize("tag", sorted(OTHER))
def test_other_tasks_episodes(oracle_mod, prec, otol, stol, tag):
    """commands, observations, rewards, dones (and info) of the recorded reference episodes"""
    n_ep = int(G[f"{tag}_n_episodes"])
    dones = 0
    for ep in range(n_ep):
        e = _other_env(oracle_mod, tag, prec)
        R = {k: G[f"{tag}_ep{ep}_{k}"] for k in ("reset_state", "obs0", "actions", "cmds", "states", "obs", "reward", "done", "info")}
        e.set_state_full(np.append(R["reset_state"], [0.0, 0.0]))
        e.set_scalar(0)
        assert np.max(np.abs(e.obs_eval() - R["obs0"])) <= otol
        last = R["reset_state"]
        for t in range(len(R["reward"])):
            cm = e.cmds_eval(R["actions"][t], last[7])
            assert np.allclose(cm, R["cmds"][t], rtol=0, atol=3e-6), (tag, ep, t, cm, R["cmds"][t])
            e.set_state_full(np.append(R["states"][t], [0.0, 0.0]))
            assert np.max(np.abs(e.obs_eval() - R["obs"][t])) <= otol, (tag, ep, t)
            r, d = e.reward_eval(last, R["cmds"][t], t == 0)
            assert d == bool(R["done"][t]), (tag, ep, t)
            assert abs(r - R["reward"][t]) <= (5e-5 if prec == "f32" else 1e-7), (tag, ep, t, r, R["reward"][t])
            if tag != "drib":
                info = e.task_out()["info"]
                assert np.allclose(info, R["info"][t], rtol=0, atol=3e-4 if prec == "f32" else 1e-6), (tag, ep, t, info, R["info"][t])
            last = R["states"][t]
        dones += int(R["done"][-1])
    assert dones >= n_ep - 2


@pytest.mark.parametrize("tag", sorted(OTHER))
def test_other_tasks_device_style_placement(oracle_mod, tag):
    task, kind = OTHER[tag]
    f = oracle_mod.OracleEnv(*kind).field_params()
    for env_id in range(100):
        e = oracle_mod.OracleEnv(*kind, 25, "f32")
        e.task_attach(task, 9, env_id, 0)
        e.task_reset()
        s = e.get_state()
        if tag == "drib":    # dribbling.py:187-202
            assert np.allclose(s[:2], [-0.1, 0]) and np.allclose(s[5:8], [0, 0, 180])
            assert [s[5 + 11 * k] for k in range(1, 5)] == [-0.5, -1.0, -1.5, -2.0]
        elif tag == "cont":  # contested_possession.py:203-220
            ex, ey = s[16], s[17]
            assert f[2] - 1e-6 <= ex <= f[0] / 2 - f[2] + 1e-6 and abs(ey) <= f[3] / 2 + 1e-6
            assert np.allclose(s[:2], [ex - 0.1, ey], atol=1e-6) and s[18] == 180 and np.all(s[5:8] == 0)
        else:                # pass_endurance.py:156-185
            bx, by = s[0], s[1]
            side = -1.0 if by < 0 else 1.0
            assert abs(bx) <= 1.5 + 1e-6 and abs(by) <= 1.5 + 1e-6
            assert np.allclose(s[5:8], [bx, by + 0.115 * side, 270 if side > 0 else 90], atol=1e-6)
            rx, ry, rth = s[16], s[17], s[18]
            assert abs(rx - bx) >= 1 - 1e-6 and abs(ry + by) < 1e-6
            want = np.rad2deg(np.arctan2(ry - s[6], rx - s[5]) + np.pi)
            assert abs(rth - want) < 2e-3
