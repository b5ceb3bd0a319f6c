"""Stepping calls captured into a hipGraph (torch.cuda.CUDAGraph) and replayed: policy(obs) -> env.step(actions), the
call pattern of the reference's training loop (rsoccer_gym/vss/vss_gym_base.py:72-90, README.md:116-133) with the policy on
the GPU.  Every replay must advance the engine's step counter — the key of its per-step random draws — exactly as an eager
call would; results are compared bit for bit with the CPU oracle fed the actions the captured policy produced."""
import numpy as np
import pytest

from helpers import f32_equal, mismatch_report

pytestmark = pytest.mark.gpu


def _policy(torch, obs_dim, act_dim, device, seed=11):
    g = torch.Generator().manual_seed(seed)   # fixed weights, drawn on the host: the same on every box
    w1 = (torch.randn(obs_dim, 32, generator=g) * 0.4).to(device)
    b1 = (torch.randn(32, generator=g) * 0.1).to(device)
    w2 = (torch.randn(32, act_dim, generator=g) * 0.6).to(device)
    return lambda obs: torch.tanh(torch.tanh(obs @ w1 + b1) @ w2)


def _oracles(O, env, seed, max_steps):
    refs = []
    for e in range(env.num_envs):
        r = O.OracleEnv(env.KIND, env.FIELD_TYPE, env.N_BLUE, env.N_YELLOW, 25, "f32")
        r.task_attach(env.TASK, seed, e, max_steps)
        r.task_reset()
        refs.append(r)
    return refs


def _compare(torch, env, refs, tag):
    torch.cuda.synchronize()
    t = env._t
    obs, rew = t["obs"].cpu().numpy(), t["reward"].cpu().numpy()
    term, trunc = t["terminated"].cpu().numpy(), t["truncated"].cpu().numpy()
    fin, steps = t["final_obs"].cpu().numpy(), t["steps"].cpu().numpy()
    state = env.sim.get_state_full()
    for e, r in enumerate(refs):
        o = r.task_out()
        assert f32_equal(obs[e], o["obs"]), mismatch_report(obs[e], o["obs"], f"obs env {e} {tag}")
        assert f32_equal(rew[e], o["reward"]), f"reward env {e} {tag}"
        assert term[e] == o["terminated"] and trunc[e] == o["truncated"] and steps[e] == o["steps"], f"flags env {e} {tag}"
        if o["terminated"] or o["truncated"]:
            assert f32_equal(fin[e], o["final_obs"]), f"final_obs env {e} {tag}"
        w = r.get_state_full()
        assert f32_equal(state[e], w), mismatch_report(state[e], w, f"state env {e} {tag}")


CASES = [
    # class name, layout (RSX_LAYOUT), batch, max_episode_steps, constructor kwargs
    ("VecVSSEnv", None, 64, 70, {}),
    ("VecVSSEnv", "epl", 70, 70, {}),                     # one lane per env; ragged batch
    ("VecSSLStaticDefendersEnv", None, 64, 40, {}),       # placement cache: helper workgroups read the tick's parity
    ("VecSSLStaticDefendersEnv", "epl", 64, 40, {}),
    ("VecSSLContestedPossessionEnv", None, 40, 30, {}),
    ("VecSSLScrimmageEnv", "quad", 24, 40, {"crowded": True}),   # four lanes per env
]


@pytest.mark.parametrize("cls,layout,B,max_steps,kw", CASES)
def test_captured_policy_loop_is_bit_identical_to_the_oracle(oracle_mod, monkeypatch, cls, layout, B, max_steps, kw):
    import torch
    from rsoccer_amd import vec
    if layout:
        monkeypatch.setenv("RSX_LAYOUT", layout)
    seed = 31
    env = getattr(vec, cls)(B, device=0, seed=seed, max_episode_steps=max_steps, **kw)
    if layout:
        want = {"epl": "one-lane-per-env", "quad": "four-lanes-per-env"}[layout]
        assert env.sim.task_layout() == want
    refs = _oracles(oracle_mod, env, seed, max_steps)
    policy = _policy(torch, env.sim.obs_dim, env.sim.act_dim, env.device)
    obs, _ = env.reset()
    actions = torch.zeros(B, env.sim.act_dim, device=env.device)

    def feed(tag):
        a = actions.cpu().numpy()
        for e, r in enumerate(refs):
            r.task_step(a[e])
        _compare(torch, env, refs, tag)

    with torch.no_grad():
        # a few eager steps first: the device counter has to pick up where the host counter stands
        for i in range(5):
            actions.copy_(policy(obs))
            env.step(actions)
            feed(f"eager {i}")
        env.enable_graph_capture()
        assert env.sim.task_tick() == 5
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                     # torch's warm-up convention before a capture; these are real steps
            for i in range(2):
                actions.copy_(policy(obs))
                env.step(actions)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
    # the oracle takes the same two steps: its observations are bit-identical, so the policy gives it the same actions
    for i in range(2):
        o = np.stack([r.task_out()["obs"] for r in refs]).astype(np.float32)
        a = policy(torch.from_numpy(o).to(env.device)).cpu().numpy()
        for e, r in enumerate(refs):
            r.task_step(a[e])
    _compare(torch, env, refs, "warm-up")
    assert env.sim.task_tick() == 7

    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        actions.copy_(policy(obs))
        env.step(actions)
    assert env.sim.task_tick() == 7                       # capturing enqueued nothing
    ended = 0
    for i in range(300):
        g.replay()
        feed(f"replay {i}")
        ended += int(env._t["terminated"].sum()) + int(env._t["truncated"].sum())
    assert ended > B                                      # auto-resets (and their random placements) were part of it
    assert env.sim.task_tick() == 307
    # eager calls and replays mix freely, and the counter travels in the checkpoint
    with torch.no_grad():
        actions.copy_(policy(obs)); env.step(actions)
    feed("eager after replays")
    blob = env.checkpoint()
    g.replay(); feed("replay after eager")
    g.replay(); feed("replay 2 after eager")
    env2 = getattr(vec, cls)(B, device=0, seed=seed, max_episode_steps=max_steps, **kw)   # host-keyed handle: same continuation
    env2.restore(blob)
    a_log = []
    with torch.no_grad():
        for i in range(2):
            a = policy(env2._t["obs"]).contiguous()
            env2.step(a)
            a_log.append(a)
    torch.cuda.synchronize()
    assert torch.equal(env2._t["obs"], env._t["obs"]) and torch.equal(env2.state, env.state)
    got = env.metrics()
    want = sum(r.task_out()["metrics"] for r in refs)
    assert got["env_steps"] == int(want[0]) and got["episodes"] == int(want[1])
    env.close(); env2.close()


def test_capture_without_enable_is_refused_loudly():
    import torch
    from rsoccer_amd import _lib, vec
    env = vec.VecVSSEnv(64, device=0, seed=3)
    obs, _ = env.reset()
    env.step(None)
    torch.cuda.synchronize()
    before = env.state.clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with pytest.raises(_lib.RsxError, match="rsx_task_enable_capture"):
        with torch.cuda.graph(g, stream=side):
            env.step(None)
    torch.cuda.synchronize()
    assert torch.equal(env.state, before)                 # nothing ran, nothing was enqueued
    assert env.sim.task_tick() == 1
    env.step(None)                                        # the handle is still good
    torch.cuda.synchronize()
    assert env.sim.task_tick() == 2
    # enabling inside a capture is refused too (the write would be replayed)
    with pytest.raises(_lib.RsxError, match="BEFORE the capture"):
        with torch.cuda.graph(torch.cuda.CUDAGraph(), stream=side):
            env.enable_graph_capture()
    env.close()
    # and so is the double-buffered raw step: which buffer is current is host state
    sim = _lib.Sim(0, 0, 3, 3, 25, 8)
    sim.state_buffers()
    g2 = torch.cuda.CUDAGraph()
    with pytest.raises(_lib.RsxError, match="cannot be captured"):
        with torch.cuda.graph(g2, stream=side):
            sim.step_dev_flip(side.cuda_stream)
    sim.step_dev_flip(side.cuda_stream)               # eagerly it works
    torch.cuda.synchronize()
    sim.close()


@pytest.mark.parametrize("cls,layout,B", [("VecVSSEnv", None, 96), ("VecSSLStaticDefendersEnv", None, 50), ("VecSSLScrimmageEnv", "quad", 20)])
def test_captured_random_action_steps_and_rollouts(oracle_mod, monkeypatch, cls, layout, B):
    """step_random (n launches) and the one-launch rollout inside a graph: n ticks per replay; the rollout of a handle with
    placement helpers re-syncs the helpers' counter slots."""
    import torch
    from rsoccer_amd import vec
    if layout:
        monkeypatch.setenv("RSX_LAYOUT", layout)
    seed, max_steps = 5, 25
    env = getattr(vec, cls)(B, device=0, seed=seed, max_episode_steps=max_steps)
    refs = _oracles(oracle_mod, env, seed, max_steps)
    env.reset()
    env.enable_graph_capture()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step_random(3)
        env.step_random(4, fused=True)
        env.step(None)
    for i in range(12):
        g.replay()
        for r in refs:
            for _ in range(8):
                r.task_step(None)
        _compare(torch, env, refs, f"replay {i}")
    assert env.sim.task_tick() == 96
    env.step_random(5, fused=True)
    env.step(None)
    for r in refs:
        for _ in range(6):
            r.task_step(None)
    _compare(torch, env, refs, "eager tail")
    env.close()


def test_device_counter_refuses_to_wrap():
    """2^32 - 1 steps per handle, enforced on the device for device-keyed handles: the launch that would wrap changes nothing."""
    import torch
    from rsoccer_amd import _lib, vec
    env = vec.VecVSSEnv(64, device=0, seed=9)
    env.reset()
    env.step(None)
    blob = env.checkpoint()
    blob[76:80].view(np.uint32)[0] = 0xFFFFFFFE           # the step counter of the checkpoint header
    env.enable_graph_capture()
    env.restore(blob)
    assert env.sim.task_tick() == 0xFFFFFFFE
    env.step(None)                                        # the last step a handle may take
    torch.cuda.synchronize()
    assert env.sim.task_tick() == 0xFFFFFFFF
    before = env.state.clone(); steps_before = env._t["steps"].clone()
    env.step(None)                                        # refused on the device
    torch.cuda.synchronize()
    assert torch.equal(env.state, before) and torch.equal(env._t["steps"], steps_before)
    with pytest.raises(_lib.RsxError, match="exhausted"):
        env.sim.task_tick()
    with pytest.raises(_lib.RsxError, match="exhausted"):
        env.metrics()
    env.close()


def test_a_hook_written_task_replays_from_a_graph_like_it_steps_eagerly():
    """The batched-hook layer (VecVSSBaseEnv: the reference's four hooks over [B] tensors — code the engine has never seen) captured
    as ONE graph: policy -> hooks -> raw step -> reward / done -> TimeLimit -> device-side auto-reset.  After enable_graph_capture()
    the frame buffers keep their roles (the previous frame by a copy instead of the buffer flip, which refuses to be captured);
    60 replays give the state, observations, rewards and episode counters of 60 eager steps, bit for bit."""
    import torch
    from rsoccer_amd.vec import VecVSSBaseEnv

    class Task(VecVSSBaseEnv):
        def __init__(self, n):
            super().__init__(0, 3, 3, 0.025, n, max_episode_steps=9)
            self.count = torch.zeros((), dtype=torch.int64, device="cuda")     # placements are a function of this device counter

        def _get_commands(self, action):
            v = torch.clamp(action * self.max_v, -self.max_v, self.max_v) / self.field.rbt_wheel_radius
            self.commands[0, 0].copy_(v[:, 0]); self.commands[0, 1].copy_(v[:, 1])

        def _frame_to_observations(self):
            f = self.frame
            return torch.stack([self.norm_pos(f.ball.x), self.norm_pos(f.ball.y), self.norm_pos(f.robots_blue[0].x),
                                self.norm_pos(f.robots_blue[0].y), torch.sin(torch.deg2rad(f.robots_blue[0].theta))], 1)

        def _calculate_reward_and_done(self):
            return self.frame.ball.x - self.last_frame.ball.x, self.frame.robots_blue[0].x > 0.2

        def _get_initial_positions(self):
            B = self.num_envs
            self.count += 1
            e = torch.arange(B, device="cuda", dtype=torch.float32)
            jitter = torch.sin(e * 12.9898 + self.count.to(torch.float32) * 78.233) * 0.15
            ball = torch.zeros(B, 4, device="cuda"); ball[:, 0] = jitter; ball[:, 1] = -jitter
            blue = torch.zeros(B, 3, 3, device="cuda"); yellow = torch.zeros(B, 3, 3, device="cuda")
            for k in range(3):
                blue[:, k, 0] = -0.5; blue[:, k, 1] = 0.3 * (k - 1)
                yellow[:, k, 0] = 0.5; yellow[:, k, 1] = 0.3 * (k - 1); yellow[:, k, 2] = 180.0
            return ball, blue, yellow

    B = 192
    w = torch.tensor([[0.9, -0.4], [0.3, 0.8], [-0.7, 0.2], [0.5, 0.5], [0.1, -0.9]], device="cuda")
    policy = lambda o: torch.tanh(o @ w + 0.4)

    def eager():
        env = Task(B)
        obs, _ = env.reset()
        ret = torch.zeros(B, device="cuda"); ends = torch.zeros(B, dtype=torch.int64, device="cuda")
        for _ in range(63):
            obs, rew, done, trunc, info = env.step(policy(obs))
            ret += rew; ends += (done | trunc).to(torch.int64)
        torch.cuda.synchronize()
        out = (env.sim.get_state_full(), obs.cpu().numpy(), ret.cpu().numpy(), ends.cpu().numpy(), env.steps.cpu().numpy())
        env.close()
        return out

    want = eager()
    env = Task(B)
    obs0, _ = env.reset()
    env.enable_graph_capture()
    obs = obs0.clone(); ret = torch.zeros(B, device="cuda"); ends = torch.zeros(B, dtype=torch.int64, device="cuda")
    for _ in range(3):                                                  # eager steps in graph mode first (and torch's warm-up)
        o, rew, done, trunc, info = env.step(policy(obs))
        obs.copy_(o); ret += rew; ends += (done | trunc).to(torch.int64)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        o, rew, done, trunc, info = env.step(policy(obs))
        obs.copy_(o); ret += rew; ends += (done | trunc).to(torch.int64)
    for _ in range(60):
        g.replay()
    torch.cuda.synchronize()
    got = (env.sim.get_state_full(), obs.cpu().numpy(), ret.cpu().numpy(), ends.cpu().numpy(), env.steps.cpu().numpy())
    for a, b, name in zip(got, want, ("state", "obs", "return", "episode ends", "steps")):
        assert np.array_equal(a, b), name
    assert want[3].sum() > B                                            # episodes did end (TimeLimit 9, the done hook) and re-start
    env.close()
