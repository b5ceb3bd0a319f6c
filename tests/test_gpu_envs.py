"""The Python surfaces on the real engine (MI355X): robosim-compatible classes, the
reference-shaped single-env tasks, the fused batched envs and the batched-hook base classes."""
import os
import random
import sys

import numpy as np
import pytest

import fake_robosim
from helpers import f32_equal

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))


def test_robosim_classes_match_the_oracle_bitwise(oracle_mod):
    from rsoccer_amd import robosim
    rng = np.random.default_rng(0)
    for cls, kind, ft, nb, ny, C in ((robosim.VSS, 0, 0, 3, 3, 2), (robosim.SSL, 1, 2, 1, 6, 8)):
        line = lambda n, s: [[s * 0.2 * i, 0, 0] for i in range(1, n + 1)]
        sim = cls(ft, nb, ny, 25, [0, 0, 0.3, 0.1], line(nb, -1), line(ny, 1))
        ref = oracle_mod.OracleEnv(kind, ft, nb, ny, 25, "f32")
        ref.reset([0, 0, 0.3, 0.1], np.array(line(nb, -1)), np.array(line(ny, 1)))
        assert set(sim.get_field_params()) == set(fake_robosim.VSS(0, 1, 0, 25, [0] * 4, [[0, 0, 0]], []).get_field_params())
        for t in range(30):
            cmds = rng.uniform(-40, 40, (nb + ny, C))
            if kind == 1:
                cmds[:, 0] = 0; cmds[:, 4:] = 0
            sim.step(cmds); ref.step(cmds)
            st = sim.get_state()
            assert st.dtype == np.float64 and f32_equal(st, ref.get_state())
        sim.close()


def test_single_env_tasks_replay_reference_episodes_on_gpu():
    """The reference-shaped VSSEnv / StaticDefenders classes with the HIP engine behind them
    reproduce the recorded reference episodes (float64 oracle physics) to fp32 accuracy."""
    from rsoccer_amd.ssl.ssl_hw_challenge import SSLHWStaticDefendersEnv
    from rsoccer_amd.vss.env_vss import VSSEnv
    env = VSSEnv()
    random.seed(102); np.random.seed(202)
    obs, _ = env.reset()
    assert np.allclose(obs, G["vss_ep2_obs0"], atol=1e-6)
    for t in range(25):
        o, r, d, tr, info = env.step(G["vss_ep2_actions"][t])
        assert np.allclose(o, G["vss_ep2_obs"][t], atol=2e-4), t
        assert abs(r - G["vss_ep2_reward"][t]) < 2e-3 and d == bool(G["vss_ep2_done"][t])
    env.close()
    env = SSLHWStaticDefendersEnv(field_type=2)
    random.seed(300)
    obs, _ = env.reset()
    assert np.allclose(obs, G["sd_ep0_obs0"], atol=1e-6)
    for t in range(20):
        o, r, d, tr, info = env.step(G["sd_ep0_actions"][t])
        assert np.allclose(o, G["sd_ep0_obs"][t], atol=2e-4), t
        assert abs(r - G["sd_ep0_reward"][t]) < 1e-3
    env.close()


def test_make_vss_v0_runs_1200_steps():
    """BASELINE.json configs[0] on the real engine: VSS-v0, 1 env, random actions."""
    import rsoccer_amd
    env = rsoccer_amd.make("VSS-v0")
    random.seed(1); np.random.seed(1)
    obs, _ = env.reset(seed=0)
    n = 0
    while True:
        obs, r, term, trunc, info = env.step(env.action_space.sample())
        n += 1
        assert obs.shape == (40,) and obs.dtype == np.float32 and np.all(np.abs(obs) <= 1.2 + 1e-6)
        if term or trunc:
            break
    assert n <= 1200 and (term or n == 1200)
    env.close()


def test_fused_observation_matches_reference_arithmetic():
    """The device-side observation of arbitrary states equals the reference's (golden vectors).  The states are written behind the
    task's back (rsx_set_state) into a handle with time step 0 — a step() there is commands -> observation without physics — and one
    step brings the observation out (`final_obs` for the states that end an episode: their `obs` is already the next episode's)."""
    import torch
    from rsoccer_amd import _lib as L
    for task, kind, ft, nb, ny, key in ((1, 0, 0, 3, 3, "vss"), (2, 1, 2, 1, 6, "sd")):
        states = G[f"{key}_obs_states"]
        B = len(states)
        sim = L.Sim(kind, ft, nb, ny, 0, B)
        sim.task_attach(task, 0, 0, 0)
        sim.task_reset()
        sim.set_state(np.concatenate([states, np.zeros((B, 2))], 1))
        sim.task_step(None)
        torch.cuda.synchronize()
        t = sim.task_tensors()
        ended = (t["terminated"].cpu().numpy() | t["truncated"].cpu().numpy()).astype(bool)
        obs = np.where(ended[:, None], t["final_obs"].cpu().numpy(), t["obs"].cpu().numpy())
        assert np.max(np.abs(obs - G[f"{key}_obs"])) <= 3e-6
        sim.close()


def test_vec_env_api_and_determinism():
    import torch
    from rsoccer_amd import vec
    for cls, od, ad in ((vec.VecVSSEnv, 40, 2), (vec.VecSSLStaticDefendersEnv, 24, 5), (vec.VecSSLDribblingEnv, 21, 4),
                        (vec.VecSSLContestedPossessionEnv, 14, 5), (vec.VecSSLPassEnduranceEnv, 16, 3)):
        a = cls(256, seed=3)
        b = cls(256, seed=3)
        c = cls(256, seed=4)
        oa, _ = a.reset(); ob, _ = b.reset(); oc, _ = c.reset()
        assert oa.shape == (256, od) and oa.dtype == torch.float32 and oa.is_cuda
        assert torch.equal(oa, ob)
        if cls is not vec.VecSSLDribblingEnv:       # the dribbling course is a fixed placement
            assert not torch.equal(oa, oc)
        gen = torch.Generator(device="cuda").manual_seed(0)
        ret = torch.zeros(256, device="cuda")
        for t in range(50):
            act = torch.rand(256, ad, device="cuda", generator=gen) * 2 - 1
            oa, ra, ta, tra, ia = a.step(act)
            ob, rb, tb, trb, ib = b.step(act.cpu().numpy())       # host actions take the staging path
            assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(ta, tb)
            assert ra.shape == (256,) and ta.dtype == torch.uint8 and set(ia) >= set(cls.INFO_KEYS)
            ret += ra
        assert torch.isfinite(ret).all() and oa.abs().max() <= 1.2 + 1e-6
        a.step_random(30); a.step_random(30, fused=True)
        m = a.metrics()
        assert m["env_steps"] == 256 * 110 and m["episodes"] >= 0
        with pytest.raises(ValueError):
            a.step(torch.zeros(3, ad, device="cuda"))
        for e in (a, b, c):
            e.close()


def test_step_async_wait_is_the_same_step():
    import torch
    from rsoccer_amd.vec import VecVSSEnv
    a = VecVSSEnv(16, seed=5)
    b = VecVSSEnv(16, seed=5)
    a.reset(); b.reset()
    act = torch.rand(16, 2, device="cuda") * 2 - 1
    for _ in range(5):
        o1, r1, t1, u1, _ = a.step(act)
        b.step_async(act)
        o2, r2, t2, u2, _ = b.step_wait(synchronize=True)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(t1, t2) and torch.equal(u1, u2)
    with pytest.raises(RuntimeError):
        b.step_wait()
    a.close(); b.close()


def test_time_limit_truncates_and_auto_resets():
    import torch
    from rsoccer_amd.vec import VecVSSEnv
    env = VecVSSEnv(64, seed=1, max_episode_steps=7)
    env.reset()
    for t in range(1, 22):
        obs, r, term, trunc, info = env.step(None)
        steps = info["episode_steps"]
        if t % 7 == 0:
            assert bool(((trunc == 1) | (term == 1)).all())
            assert bool((steps == 0).all())
            assert not torch.equal(obs, info["final_obs"])
        else:
            ended = (term == 1)
            assert bool((trunc[~ended] == 0).all())
    m = env.metrics()
    assert m["episodes"] >= 64 * 3 and m["episode_len_sum"] <= 64 * 21
    env.close()


def test_batched_hooks_env_matches_fused_kernel():
    """A VSS-v0-like task written with the batched hooks (torch on VecFrame) sees the same
    physics as the raw kernel and produces the reference observation layout."""
    import torch
    from rsoccer_amd.vec import VecVSSBaseEnv

    class Task(VecVSSBaseEnv):
        def __init__(self, n):
            super().__init__(0, 3, 3, 0.025, n, auto_reset=False)
            self.rng = np.random.default_rng(0)

        def _get_commands(self, action):
            v = torch.clamp(action * self.max_v, -self.max_v, self.max_v)
            v = torch.where(v.abs() < 0.05, torch.zeros_like(v), v) / self.field.rbt_wheel_radius
            self.commands[0, 0].copy_(v[:, 0]); self.commands[0, 1].copy_(v[:, 1])

        def _frame_to_observations(self):
            f = self.frame
            cols = [self.norm_pos(f.ball.x), self.norm_pos(f.ball.y), self.norm_v(f.ball.v_x), self.norm_v(f.ball.v_y)]
            for i in range(3):
                r = f.robots_blue[i]
                th = torch.deg2rad(r.theta)
                cols += [self.norm_pos(r.x), self.norm_pos(r.y), torch.sin(th), torch.cos(th), self.norm_v(r.v_x),
                         self.norm_v(r.v_y), self.norm_w(r.v_theta)]
            for i in range(3):
                r = f.robots_yellow[i]
                cols += [self.norm_pos(r.x), self.norm_pos(r.y), self.norm_v(r.v_x), self.norm_v(r.v_y), self.norm_w(r.v_theta)]
            return torch.stack(cols, 1)

        def _calculate_reward_and_done(self):
            done = self.frame.ball.x.abs() > self.field.length / 2
            moved = self.frame.ball.x - self.last_frame.ball.x
            return moved, done

        def _get_initial_positions(self):
            B = self.num_envs
            ball = np.zeros((B, 4)); ball[:, :2] = self.rng.uniform(-0.3, 0.3, (B, 2))
            blue = np.zeros((B, 3, 3)); yellow = np.zeros((B, 3, 3))
            for k in range(3):
                blue[:, k, :2] = [-0.5, 0.3 * (k - 1)]
                yellow[:, k, :2] = [0.5, 0.3 * (k - 1)]; yellow[:, k, 2] = 180.0
            return ball, blue, yellow

    env = Task(128)
    obs, _ = env.reset()
    assert obs.shape == (128, 40) and obs.is_cuda
    from oracle import oracle as O
    ref = O.OracleEnv(0, 0, 3, 3, 25, "f32")
    st0 = env.sim.get_state_full()
    ref.set_state_full(st0[5])
    act = torch.full((128, 2), 0.5, device="cuda")
    for _ in range(10):
        obs, rew, done, trunc, _ = env.step(act)
        ref.step(env.commands[:, :, 5].cpu().numpy().astype(np.float64))
    torch.cuda.synchronize()
    assert f32_equal(env.sim.get_state_full()[5], ref.get_state_full())
    assert rew.shape == (128,) and done.dtype == torch.bool
    assert float(env.frame.robots_blue[0].v_x.mean()) > 0.1      # the agent robot did move
    assert float(env.frame.robots_yellow[1].v_x.abs().max()) == 0.0
    env.close()


def _hook_task(n, device_placement, **kw):
    import torch
    from rsoccer_amd.vec import VecVSSBaseEnv

    class Task(VecVSSBaseEnv):
        """VSS-v0-shaped task written with the batched hooks"""

        def __init__(self):
            super().__init__(0, 3, 3, 0.025, n, **kw)
            g = torch.Generator(device="cuda"); g.manual_seed(3)
            self.gen = g

        def _get_commands(self, action):
            v = torch.clamp(action * self.max_v, -self.max_v, self.max_v) / self.field.rbt_wheel_radius
            self.commands[0, 0].copy_(v[:, 0]); self.commands[0, 1].copy_(v[:, 1])

        def _frame_to_observations(self):
            f = self.frame
            cols = [self.norm_pos(f.ball.x), self.norm_pos(f.ball.y)]
            for i in range(3):
                cols += [self.norm_pos(f.robots_blue[i].x), self.norm_pos(f.robots_blue[i].y)]
            return torch.stack(cols, 1)

        def _calculate_reward_and_done(self):
            return self.frame.ball.x - self.last_frame.ball.x, self.frame.robots_blue[0].x > -0.45

        def _get_initial_positions(self):
            B = self.num_envs
            ball = torch.zeros(B, 4, device="cuda"); ball[:, :2] = torch.rand(B, 2, device="cuda", generator=self.gen) * 0.4 - 0.2
            blue = torch.zeros(B, 3, 3, device="cuda"); yellow = torch.zeros(B, 3, 3, device="cuda")
            for k in range(3):
                blue[:, k, 0] = -0.5; blue[:, k, 1] = 0.3 * (k - 1)
                yellow[:, k, 0] = 0.5; yellow[:, k, 1] = 0.3 * (k - 1); yellow[:, k, 2] = 180.0
            if device_placement:
                return ball, blue, yellow
            return ball.cpu().numpy().astype(np.float64), blue.cpu().numpy().astype(np.float64), yellow.cpu().numpy().astype(np.float64)
    return Task()


def test_batched_hooks_timelimit_and_auto_reset_stay_on_the_device():
    """_VecBaseEnv: TimeLimit (`truncated` is really set), same-step auto-reset from DEVICE placements,
    last_frame from the second state buffer — and no host<->device copy or synchronisation in step()
    (torch's sync-debug mode turns any into an error; the engine calls are launches only)."""
    import torch
    env = _hook_task(256, True, max_episode_steps=7)
    obs, _ = env.reset()
    act = torch.zeros(256, 2, device="cuda"); act[:128] = 1.0          # half of the envs drive forward
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        log = []
        for t in range(16):
            prev_ball = env.frame.ball.x.clone()
            obs, rew, done, trunc, info = env.step(act)
            log.append((obs, rew, done, trunc, info["final_obs"], env.steps.clone(), prev_ball, env.last_frame.ball.x.clone()))
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    for t, (obs, rew, done, trunc, fin, steps, prev_ball, last_ball) in enumerate(log):
        assert torch.equal(last_ball, prev_ball)                      # last_frame IS the pre-step state
        ended = (done | trunc).cpu().numpy()
        if t == 6:                                                    # TimeLimit: every env not already re-started
            assert trunc.cpu().numpy()[128:].all()
        assert (steps.cpu().numpy()[ended] == 0).all() and (steps.cpu().numpy()[~ended] > 0).all()
        o, f = obs.cpu().numpy(), fin.cpu().numpy()
        assert np.array_equal(o[~ended], f[~ended])
        if ended.any():                                               # re-placed: robots back on their marks
            assert np.allclose(o[ended][:, 2], env.norm_pos(torch.tensor(-0.5)).item())
            assert not np.array_equal(o[ended], f[ended])
    assert any(l[2].any().item() for l in log) and any(l[3].any().item() for l in log)
    env.close()
    # host-array placements still work (robosim.reset format), with copies
    env = _hook_task(32, False, max_episode_steps=5)
    env.reset()
    for _ in range(6):
        obs, rew, done, trunc, info = env.step(torch.zeros(32, 2, device="cuda"))
    assert int(env.steps.max().item()) == 1
    env.close()


def test_scrimmage_env_11v11_api():
    """VecSSLScrimmageEnv: BASELINE configs[3] as a fused task (11v11 on the division-A field, every robot
    commanded): shapes, fed and device-drawn actions, goals end episodes, crowded line-up is a scrum."""
    import torch
    from rsoccer_amd.vec import VecSSLScrimmageEnv
    env = VecSSLScrimmageEnv(256, seed=4)
    obs, _ = env.reset()
    assert obs.shape == (256, 46) and env.single_action_space.shape == (88,)
    x = env.state[5::11][:22]
    assert float((x[:, None] - x[None]).abs().add(torch.eye(22, device="cuda")[:, :, None] * 9).min()) >= 0.0   # placed
    for _ in range(30):
        obs, rew, term, trunc, info = env.step(torch.rand(256, 88, device="cuda") * 2 - 1)
    obs, rew, term, trunc, info = env.step_random(400)
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and obs.abs().max() <= 1.2
    m = env.metrics()
    assert m["env_steps"] == 256 * 430 and m["goals_for"] + m["goals_against"] == m["episodes"] - m["truncated_episodes"]
    env.close()
    crowded = VecSSLScrimmageEnv(64, crowded=True, seed=4)
    crowded.reset()
    st = crowded.state
    assert float(st[5::11][:22].abs().max()) < 0.8 and float(st[6::11][:22].abs().max()) < 0.5    # everyone within the 1.5 m disc
    crowded.step_random(50)
    assert crowded.sim.check_finite() == 0
    crowded.close()


def test_scalar_hook_adapter_equals_independent_single_envs():
    """VecScalarHookEnv: unmodified reference-shaped task classes (scalar hooks, vss_gym_base.py:197-211)
    as slots of ONE batched simulator give exactly what the same classes give as independent
    single-env objects (same global random streams, same call order)."""
    from rsoccer_amd.ssl.ssl_hw_challenge import SSLHWStaticDefendersEnv
    from rsoccer_amd.vec import VecScalarHookEnv
    from rsoccer_amd.vss.env_vss import VSSEnv
    for cls, kw, T, limit in ((VSSEnv, {}, 60, 25), (SSLHWStaticDefendersEnv, {"field_type": 2}, 40, 15)):
        B = 6
        arng = np.random.default_rng(1)
        adim = cls(**kw).action_space.shape[0]
        actions = arng.uniform(-1, 1, (T, B, adim)).astype(np.float32)
        random.seed(9); np.random.seed(9)
        vec = VecScalarHookEnv(cls, B, max_episode_steps=limit, **kw)
        o0, _ = vec.reset()
        got = [vec.step(actions[t]) for t in range(T)]
        vec.close()
        random.seed(9); np.random.seed(9)
        envs = [cls(**kw) for _ in range(B)]
        obs = np.stack([e.reset()[0] for e in envs])
        assert np.array_equal(obs, o0)
        el = np.zeros(B, int)
        for t in range(T):
            outs = [e.step(actions[t, i].copy()) for i, e in enumerate(envs)]   # NB: np.random order == adapter's (OU per env, in order)
            obs = np.stack([o[0] for o in outs]); rew = np.array([o[1] for o in outs]); term = np.array([bool(o[2]) for o in outs])
            el += 1
            trunc = el >= limit
            assert np.array_equal(got[t][4]["final_obs"], obs) and np.array_equal(got[t][1], rew), (cls.__name__, t)
            assert np.array_equal(got[t][2], term) and np.array_equal(got[t][3], trunc)
            for i in np.nonzero(term | trunc)[0]:
                obs[i] = envs[i].reset()[0]; el[i] = 0
            assert np.array_equal(got[t][0], obs)
        for e in envs:
            e.close()


@pytest.mark.parametrize("env_id", ["VSS-v0", "SSLStaticDefenders-v0", "SSLDribbling-v0",
                                    "SSLContestedPossession-v0", "SSLPassEndurance-v0"])
def test_every_registered_id_steps_on_the_engine(env_id):
    """the five ids of the reference registry, constructed through make(), physics on the GPU"""
    import rsoccer_amd
    env = rsoccer_amd.make(env_id)
    random.seed(5); np.random.seed(5)
    obs, info = env.reset()
    assert obs.shape == env.observation_space.shape and obs.dtype == np.float32
    for _ in range(60):
        obs, r, term, trunc, info = env.step(env.action_space.sample())
        assert np.all(np.isfinite(obs)) and np.isfinite(r)
        if term or trunc:
            obs, info = env.reset()
    env.close()


def test_pass_endurance_ball_is_held_and_can_be_passed():
    """pass_endurance.py:156-185 places the ball 0.115 m in front of the shooter and expects the
    dribbler to hold it and the kicker to release it."""
    from rsoccer_amd.ssl.ssl_hw_challenge import SSLPassEnduranceEnv
    env = SSLPassEnduranceEnv()
    random.seed(11)
    env.reset()
    for _ in range(5):
        obs, r, d, _, _ = env.step(np.array([0.0, 0.0, 1.0], dtype=np.float32))
    assert env.frame.robots_blue[0].infrared          # shooter holds the ball
    b0 = (env.frame.ball.x, env.frame.ball.y)
    obs, r, d, _, _ = env.step(np.array([0.0, 1.0, 1.0], dtype=np.float32))   # kick
    obs, r, d, _, _ = env.step(np.array([0.0, 0.0, 0.0], dtype=np.float32))
    assert np.hypot(env.frame.ball.v_x, env.frame.ball.v_y) > 3.0
    assert np.hypot(env.frame.ball.x - b0[0], env.frame.ball.y - b0[1]) > 0.1
    env.close()


def test_plain_c_host_drives_the_library(tmp_path):
    """examples/rsx_c_host.c: dlopen + the C-ABI from a C program (no Python, no torch in that
    process) — robosim-style reset / step / get_state and a fused VSS-v0 run."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "rsx_c_host"
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "rsx_c_host.c"), "-o", str(exe), "-ldl"])
    res = subprocess.run([str(exe), os.path.join(root, "rsoccer_amd", "librsx_hip.so")], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "ok" in res.stdout


def test_vec_env_checkpoint_restore_continues_the_same_run():
    """VecVSSEnv.checkpoint() / restore(): a second env object (same seed) picks the run up mid-episode — policy
    actions, OU noise, auto-resets and counters continue as if nothing had happened"""
    import torch
    from rsoccer_amd.vec import VecVSSEnv
    a = VecVSSEnv(48, seed=9, max_episode_steps=25)
    a.reset()
    g = torch.Generator(device="cpu").manual_seed(3)
    acts = [torch.rand(48, 2, generator=g) * 2 - 1 for _ in range(60)]
    for t in range(20):
        a.step(acts[t].cuda())
    blob = a.checkpoint()
    outs = [tuple(x.clone() for x in a.step(acts[t].cuda())[:4]) for t in range(20, 60)]
    ma = a.metrics()
    b = VecVSSEnv(48, seed=9, max_episode_steps=25)
    obs, _ = b.restore(blob.tobytes())          # bytes work as well as the numpy blob
    for t in range(20, 60):
        got = b.step(acts[t].cuda())[:4]
        for x, y in zip(got, outs[t - 20]):
            assert torch.equal(x, y), t
    assert b.metrics() == ma and ma["episodes"] >= 48 * 2
    a.close(); b.close()


@pytest.mark.parametrize("name,capture", [("VecVSSEnv", False), ("VecSSLStaticDefendersEnv", False), ("VecSSLContestedPossessionEnv", True)])
def test_vec_env_reset_with_a_seed_starts_over_like_a_fresh_env(name, capture):
    """``reset(seed=s)`` of a fused vector env (gymnasium's way of seeding, README.md:116-133): the handle starts over as if it had just
    been constructed with seed ``s`` (rsx_task_reseed) — placements, OU noise, random actions, counters, metrics; ``reset()`` without a
    seed goes on with the streams it has.  Also on a handle switched to device-keyed stepping (graph capture)."""
    import torch
    from rsoccer_amd import vec
    cls = getattr(vec, name)
    B = 300
    used = cls(B, seed=5)
    if capture:
        used.enable_graph_capture()
    used.reset()
    used.step_random(37)                                  # some history under another seed
    torch.cuda.synchronize()
    assert used.metrics()["env_steps"] == B * 37
    o_used, _ = used.reset(seed=11)
    fresh = cls(B, seed=11)
    if capture:
        fresh.enable_graph_capture()
    o_fresh, _ = fresh.reset()
    assert torch.equal(o_used, o_fresh)
    for _ in range(3):
        used.step_random(20); fresh.step_random(20)
        a = torch.rand(B, used.sim.act_dim, device="cuda") * 2 - 1
        ou, ru, tu, tru, _ = used.step(a)
        of, rf, tf, trf, _ = fresh.step(a)
        assert torch.equal(ou, of) and torch.equal(ru, rf) and torch.equal(tu, tf) and torch.equal(tru, trf)
    torch.cuda.synchronize()
    assert np.array_equal(used.sim.get_state_full(), fresh.sim.get_state_full())
    assert used.metrics() == fresh.metrics()
    assert used.sim.task_tick() == fresh.sim.task_tick() == 63
    # without a seed the run goes on: a second env on the same seed that is reset() twice differs from one reset(seed=) twice
    o2, _ = used.reset()
    o3, _ = fresh.reset(seed=11)
    assert not torch.equal(o2, o3) or name == "VecSSLDribblingEnv"
    used.close(); fresh.close()


def test_fused_policy_example_kernel_matches_the_torch_policy_and_replays_from_a_graph():
    """examples/fused_policy.hip (the 40-64-2 tanh MLP of examples/vec_policy_loop.py as ONE hand-written kernel): same actions as the
    four-kernel torch form, also for a batch that is not a multiple of the envs a wave serves and for the SSLStaticDefenders shapes;
    policy -> env.step captured into a hipGraph replays like the eager calls."""
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import vec_policy_loop as VPL
    from rsoccer_amd import vec
    for od, ad, B in ((40, 2, 4096), (40, 2, 1023), (24, 5, 770)):
        ref, fused = VPL.make_policy(od, ad, torch.device("cuda", 0)), VPL.make_fused_policy(od, ad, torch.device("cuda", 0))
        obs = torch.empty(B, od, device="cuda").uniform_(-1.2, 1.2)
        want, got = ref(obs, torch.empty(B, ad, device="cuda")), fused(obs, torch.full((B, ad), 7.0, device="cuda"))
        torch.cuda.synchronize()
        assert float((got - want).abs().max()) <= 2e-5 and float(got.abs().max()) <= 1.0
    def run(graph):
        env = vec.VecVSSEnv(512, device=0, seed=5)
        env.reset()
        pol = VPL.make_fused_policy(env.sim.obs_dim, env.sim.act_dim, env.device)
        actions = torch.zeros(512, 2, device="cuda")
        with torch.no_grad():
            if graph:
                g = VPL.build_graph(env, pol, actions, 4)     # (takes one eager step itself: torch's warm-up convention)
                for _ in range(10):
                    g.replay()
            else:
                VPL.run_eager(env, pol, actions, 41)
        torch.cuda.synchronize()
        out = (env.sim.get_state_full(), env._t["obs"].cpu().numpy().copy(), env.metrics())
        env.close()
        return out
    a, b = run(False), run(True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]


def test_make_vec_builds_the_fused_env_of_every_registered_id():
    """rsoccer_amd.make_vec(id, num_envs): the registry's episode limit by default, keywords reach the class, the env steps"""
    import json
    import torch
    import rsoccer_amd
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "registry.json")))
    for env_id, spec in ref.items():
        env = rsoccer_amd.make_vec(env_id, 96, seed=3, env_id_base=1000)
        assert env.num_envs == 96 and env.max_episode_steps == spec["max_episode_steps"]
        obs, info = env.reset()
        assert obs.shape == (96, env.single_observation_space.shape[0]) and info == {}
        obs, rew, term, trunc, info = env.step(None)
        torch.cuda.synchronize()
        assert torch.isfinite(obs).all() and rew.shape == (96,)
        env.close()
    env = rsoccer_amd.make_vec("VSS-v0", 8, max_episode_steps=5)
    assert env.max_episode_steps == 5
    env.close()
