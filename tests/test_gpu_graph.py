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
