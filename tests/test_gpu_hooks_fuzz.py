"""The batched-hook layer (rsoccer_amd/vec/hooks.py: the reference's subclass contract, rsoccer_gym/vss/vss_gym_base.py:72-106,197-211
and ssl/ssl_gym_base.py:73-106,197-211, over [B] tensors with the episode bookkeeping of gymnasium.vector on the device) against a
MODEL of what it promises, under random call sequences.

The model is numpy + one oracle env per env id: `step()` = commands -> physics -> observation / reward / done from the new frame and
the frame before it -> TimeLimit -> same-step re-placement of the envs that ended (their `obs` the first of the new episode,
`info["final_obs"]` the terminal one, their step count 0); `reset(mask)` re-places the selected envs.  Checked after every call: the
current frame, `last_frame`, observations, rewards, both flags, `final_obs`, step counts — exactly.  Configurations: VSS and SSL base
classes, device-tensor and host-array placements, with and without `last_frame`, with and without auto-reset, eager and (after
`enable_graph_capture()`) replayed from a hipGraph.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import f32_equal, mismatch_report  # noqa: E402

pytestmark = pytest.mark.gpu


def _task(base_name, n, device_placement, **kw):
    import torch
    from rsoccer_amd import vec
    base = getattr(vec, base_name)
    ssl = base_name == "VecSSLBaseEnv"
    nb, ny = (2, 1) if ssl else (3, 3)

    class Task(base):
        """a task written against the four hooks, of the kind the reference's README.md:78-110 shows"""

        def __init__(self):
            super().__init__(2 if ssl else 0, nb, ny, 0.025, n, **kw)
            # placements are a function of a DEVICE counter (a captured step() must not read anything back); the values the hook handed
            # out last stay in three persistent buffers, which is where the model takes them from
            self.calls = torch.zeros((), dtype=torch.int64, device="cuda")
            self.pl = (torch.zeros(n, 4, device="cuda"), torch.zeros(n, nb, 3, device="cuda"), torch.zeros(n, ny, 3, device="cuda"))

        def _get_commands(self, action):
            if ssl:   # local velocities for every blue robot, kick when the third action is high
                self.commands[0, 1].copy_(action[:, 0] * 2.0); self.commands[0, 2].copy_(action[:, 1] * 2.0)
                self.commands[0, 3].copy_(action[:, 2] * 8.0)
                self.commands[0, 5].copy_(torch.where(action[:, 2] > 0.5, 4.0, 0.0)); self.commands[0, 7].fill_(1.0)
                self.commands[1, 1].copy_(action[:, 1])
            else:
                v = torch.clamp(action[:, :2] * self.max_v, -self.max_v, self.max_v) / self.field.rbt_wheel_radius
                self.commands[0, 0].copy_(v[:, 0]); self.commands[0, 1].copy_(v[:, 1])
                self.commands[4, 0].copy_(v[:, 1]); self.commands[4, 1].copy_(v[:, 1])

        def _frame_to_observations(self):
            f = self.frame
            return torch.stack([self.norm_pos(f.ball.x), self.norm_pos(f.ball.y), self.norm_v(f.ball.v_x), self.norm_pos(f.robots_blue[0].x),
                                self.norm_pos(f.robots_blue[0].y), self.norm_w(f.robots_blue[0].v_theta)], 1)

        def _calculate_reward_and_done(self):
            moved = (self.frame.ball.x - self.last_frame.ball.x) if self.keep_last_frame else self.frame.ball.x * 0.0
            return moved + self.frame.robots_blue[0].v_x * 0.01, self.frame.robots_blue[0].x > (0.3 if ssl else -0.35)

        def _get_initial_positions(self):
            B = self.num_envs
            self.calls += 1
            e = torch.arange(B, device="cuda", dtype=torch.int64)
            u = [(((e * (73 + 31 * j) + self.calls * (151 + 17 * j)) % 997).to(torch.float32) / 997.0) for j in range(4)]
            ball = torch.zeros(B, 4, device="cuda"); ball[:, 0] = u[0] * 0.4 - 0.2; ball[:, 1] = u[1] * 0.4 - 0.2; ball[:, 2] = u[2] - 0.5; ball[:, 3] = u[3] - 0.5
            blue = torch.zeros(B, nb, 3, device="cuda"); yellow = torch.zeros(B, ny, 3, device="cuda")
            for k in range(nb):
                blue[:, k, 0] = -0.5 - 0.3 * k * ssl; blue[:, k, 1] = 0.3 * (k - 1); blue[:, k, 2] = u[0] * 90.0
            for k in range(ny):
                yellow[:, k, 0] = 0.5; yellow[:, k, 1] = 0.3 * (k - 1); yellow[:, k, 2] = 180.0
            self.pl[0].copy_(ball); self.pl[1].copy_(blue); self.pl[2].copy_(yellow)
            if device_placement:
                return ball, blue, yellow
            return tuple(t.cpu().numpy().astype(np.float64) for t in (ball, blue, yellow))
    return Task(), ssl, nb, ny


class _Model:
    def __init__(self, O, env, ssl, nb, ny):
        self.env, self.ssl, self.nb, self.ny = env, ssl, nb, ny
        B = env.num_envs
        self.refs = [O.OracleEnv(1 if ssl else 0, 2 if ssl else 0, nb, ny, 25, "f32") for _ in range(B)]
        self.steps = np.zeros(B, dtype=np.int64)
        self.last = None
        self.used = 0     # re-placements so far (= calls of the placement hook that executed)
        self.f32 = np.float32

    def state(self):
        return np.stack([r.get_state_full() for r in self.refs]).astype(np.float32)

    def _t(self, a):
        import torch
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def obs_of(self, st):
        """the task's observation formula on a model state — evaluated by torch on the device like the hook's (the subject here is the
        layer's bookkeeping, not whether numpy and torch round `x / scalar` alike: torch multiplies by the reciprocal)"""
        import torch
        e, t = self.env, self._t(st)
        rb = 5   # blue robot 0: x, y, theta, vx, vy, vtheta at 5..10
        return torch.stack([e.norm_pos(t[:, 0]), e.norm_pos(t[:, 1]), e.norm_v(t[:, 3]), e.norm_pos(t[:, rb]), e.norm_pos(t[:, rb + 1]),
                            e.norm_w(t[:, rb + 5])], 1).cpu().numpy()

    def place(self, mask):
        self.used += 1
        assert int(self.env.calls.item()) == self.used, "the placement hook runs exactly once per re-placement"
        ball, blue, yellow = (t.cpu().numpy().astype(np.float64) for t in self.env.pl)
        for i, r in enumerate(self.refs):
            if mask is None or mask[i]:
                r.reset(ball[i], blue[i], yellow[i] if self.ny else np.zeros(0))

    def reset(self, mask):
        self.place(mask)
        self.steps[np.ones_like(self.steps, dtype=bool) if mask is None else mask.astype(bool)] = 0
        self.last = None
        return self.obs_of(self.state())

    def step(self, cmds):
        env, f = self.env, np.float32
        self.steps += 1
        before = self.state()
        for i, r in enumerate(self.refs):
            r.step(cmds[:, :, i].astype(np.float64))
        st = self.state()
        obs = self.obs_of(st)
        ts, tb = self._t(st), self._t(before)
        moved = (ts[:, 0] - tb[:, 0]) if env.keep_last_frame else ts[:, 0] * 0.0
        reward = (moved + ts[:, 8] * 0.01).cpu().numpy()
        done = (ts[:, 5] > (0.3 if self.ssl else -0.35)).cpu().numpy()
        trunc = self.steps >= env.max_episode_steps if env.max_episode_steps else np.zeros_like(done)
        final = obs.copy()
        self.last = before
        if env.auto_reset:
            ended = done | trunc
            # device placements re-place (masked) on every step; host-array placements only when an episode ended
            if env._device_placement or ended.any():
                self.place(ended)
                self.steps[ended] = 0
                obs = np.where(ended[:, None], self.obs_of(self.state()), obs)
        return obs, reward, done, trunc, final


def _check(env, model, out, want, log):
    import torch
    torch.cuda.synchronize()
    ctx = " ".join(log[-10:])
    cur = env.sim.state_buffers()[0].cpu().numpy().T if env.keep_last_frame else env.sim.state_tensor().cpu().numpy().T
    assert f32_equal(cur, model.state()), mismatch_report(cur, model.state(), "simulator state: " + ctx)
    assert f32_equal(env.frame.state.cpu().numpy().T, model.state()), "env.frame is not the current frame: " + ctx
    assert np.array_equal(env.steps.cpu().numpy(), model.steps), (env.steps.cpu().numpy(), model.steps, ctx)
    if out is None:
        return
    if len(out) == 2:    # reset
        assert f32_equal(out[0].cpu().numpy(), want), mismatch_report(out[0].cpu().numpy(), want, "reset obs: " + ctx)
        assert env.last_frame is None or env._graph_mode    # (graph mode: the buffers keep their roles, hooks.py: reset)
        return
    obs, rew, done, trunc, info = out
    wobs, wrew, wdone, wtrunc, wfinal = want
    assert f32_equal(obs.cpu().numpy(), wobs), mismatch_report(obs.cpu().numpy(), wobs, "obs: " + ctx)
    assert f32_equal(rew.cpu().numpy(), wrew), mismatch_report(rew.cpu().numpy(), wrew, "reward: " + ctx)
    assert np.array_equal(done.cpu().numpy(), wdone) and np.array_equal(trunc.cpu().numpy().astype(bool), wtrunc.astype(bool)), "flags: " + ctx
    if env.auto_reset:
        assert f32_equal(info["final_obs"].cpu().numpy(), wfinal), "final_obs: " + ctx
    if env.keep_last_frame:
        assert f32_equal(env.last_frame.state.cpu().numpy().T, model.last), "last_frame is not the frame before the step: " + ctx


def run_hooks_sequence(O, base_name, seed, n_ops):
    import torch
    rng = np.random.default_rng(seed)
    B = int(rng.choice([5, 64, 130]))
    device_placement = bool(rng.random() < 0.6)
    kw = dict(keep_last_frame=bool(rng.random() < 0.75), auto_reset=bool(rng.random() < 0.8),
              max_episode_steps=int(rng.choice([0, 4, 9])) or None)
    env, ssl, nb, ny = _task(base_name, B, device_placement, **kw)
    model = _Model(O, env, ssl, nb, ny)
    log = [f"{base_name} B={B} device_placement={device_placement} {kw}"]
    graph = None
    act = torch.zeros(B, 3, device="cuda")
    try:
        out = env.reset()
        _check(env, model, out, model.reset(None), log)
        for _ in range(n_ops):
            op = rng.choice(["step", "step", "step", "step", "reset", "reset_mask", "graph", "replay"])
            if op == "graph" and (graph is not None or not (device_placement or not env.auto_reset)):
                op = "step"
            if op == "replay" and graph is None:
                op = "step"
            log.append(op)
            if op in ("step", "replay"):
                a = rng.uniform(-1, 1, (B, 3)).astype(np.float32)
                act.copy_(torch.from_numpy(a))
                if op == "step":
                    out = env.step(act)
                else:
                    graph[0].replay()
                    out = graph[1]
                torch.cuda.synchronize()
                cmds = env.commands.cpu().numpy()            # [n_robots, C, B]: what the hook under test wrote
                _check(env, model, out, model.step(cmds), log)
            elif op == "reset":
                out = env.reset()
                _check(env, model, out, model.reset(None), log)
            elif op == "reset_mask":
                m = rng.random(B) < 0.4
                out = env.reset(torch.from_numpy(m).cuda() if rng.random() < 0.5 else m)
                _check(env, model, out, model.reset(m), log)
            else:   # capture one step() of the hook-written task into a hipGraph; later `replay` ops run it
                env.enable_graph_capture()
                out = env.step(act)                           # an eager step in graph mode first (torch's warm-up convention: a real step)
                torch.cuda.synchronize()
                _check(env, model, out, model.step(env.commands.cpu().numpy()), log)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    gout = env.step(act)
                graph = (g, gout)
                _check(env, model, None, None, log)           # capturing executes nothing: state, counters and the hook's device counter stand
    finally:
        env.close()
    return log


@pytest.mark.parametrize("base_name", ["VecVSSBaseEnv", "VecSSLBaseEnv"])
def test_hook_layer_matches_its_model_under_random_call_sequences(oracle_mod, base_name):
    for seed in range(40, 52):
        run_hooks_sequence(oracle_mod, base_name, seed, 40)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    g.build()
    from oracle import oracle as O
    O.build()
    bad = 0
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for base_name in ("VecVSSBaseEnv", "VecSSLBaseEnv"):
        for seed in range(1000, 1000 + n):
            try:
                run_hooks_sequence(O, base_name, seed, 80)
            except AssertionError as ex:
                bad += 1
                print(f"FAIL {base_name} seed {seed}: {str(ex)[:1200]}", flush=True)
        print(f"{base_name}: {n} sequences x 80 calls, failures so far {bad}", flush=True)
    sys.exit(1 if bad else 0)
