"""Differential fuzzing of the C-ABI's CALL SEQUENCES against the CPU oracle.

The parity tests of test_gpu_parity.py exercise one feature at a time (fed / random actions, step_n, rollout, masked reset_to,
checkpoints, graphs, reseed).  A caller of the reference-shaped API interleaves them freely — the trainer of
rsoccer_gym's README.md:116-133 resets in the middle of an episode, the adapters of Simulators/rsim.py:36-110 mix reset / step /
get_state — so this file draws RANDOM interleavings of every stepping, resetting and state-carrying call of a fused handle
(rsx_task_step with fed and with device-drawn actions, _step_n, _rollout, _reset, masked _reset_to, checkpoint save -> load into a
NEW handle of another kernel layout, rsx_task_enable_capture, rsx_task_reseed, rsx_read_metrics, rsx_task_tick) and checks after
EVERY call that observations, rewards, flags, info rows, step counters, terminal observations, the full simulator state and the
device-side metrics equal what one oracle env per env id produces for the same sequence — bit for bit.

The same for the simulator surface without a task (what replaces robosim.{VSS,SSL}: host float64 and device SoA forms, zero-copy and
wire-buffer paths, poses and states far outside what a task produces), for the sequences driven through rsoccer_amd.vec, and — in the
campaign form — for random configurations (team sizes, field types, time steps, batch sizes).

`python tests/test_gpu_api_fuzz.py [--seeds N] [--ops M] [--random-configs K]` runs a longer campaign on the GPU box (what
profiles/r06_api_fuzz.txt holds).
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import f32_equal, mismatch_report, random_placement  # noqa: E402

pytestmark = pytest.mark.gpu

# task, kind, field type, n_blue, n_yellow, batch, TimeLimit, layouts the handle may be stepped by
CONFIGS = {
    "VSS-v0": (1, 0, 0, 3, 3, 70, 23, ("lanes", "epl")),
    "VSS-v0-5v5": (1, 0, 1, 5, 5, 19, 17, ("lanes",)),
    "SSLStaticDefenders-v0": (2, 1, 2, 1, 6, 66, 19, ("lanes", "epl")),
    "SSLDribbling-v0": (3, 1, 2, 1, 4, 37, 29, ("lanes", "epl")),
    "SSLContestedPossession-v0": (4, 1, 2, 1, 1, 45, 13, ("lanes", "epl")),
    "SSLPassEndurance-v0": (5, 1, 2, 2, 0, 33, 11, ("lanes", "epl")),
    "SSLPassEndurance-v0-long-episodes": (5, 1, 2, 2, 0, 17, 70, ("lanes", "epl")),   # long enough for the stalled-ball end (pass_endurance.py:205-214)
    "scrimmage-11v11-crowded": (7, 1, 1, 11, 11, 9, 15, ("lanes", "quad")),
    "scrimmage-3v2": (6, 1, 2, 3, 2, 21, 9, ("lanes",)),
}


def wild_placement(rng, B, nb, ny, f):
    """poses no task would place: anywhere up to half a metre outside the field (behind walls, inside and beside the goals) or crowded into
    a goal mouth / a corner, robots on top of each other (also EXACTLY: d2 == 0), the ball inside a robot, the ball at speed"""
    N = nb + ny
    hx, hy = f["length"] / 2 + 0.5, f["width"] / 2 + 0.5
    if rng.random() < 0.5:
        cx = rng.choice([-1, 1]) * f["length"] / 2
        cy = rng.choice([0.0, f["goal_width"] / 2, f["width"] / 2, -f["width"] / 2])
        pos = np.stack([cx + rng.normal(0, 0.12, (B, N + 1)), cy + rng.normal(0, 0.12, (B, N + 1))], -1)
    else:
        pos = np.stack([rng.uniform(-hx, hx, (B, N + 1)), rng.uniform(-hy, hy, (B, N + 1))], -1)
    for e in range(B):
        if rng.random() < 0.3:
            i, j = rng.integers(0, N + 1, 2)
            pos[e, i] = pos[e, j]
    ball = np.concatenate([pos[:, 0], rng.uniform(-6, 6, (B, 2))], -1)
    rob = np.concatenate([pos[:, 1:], rng.uniform(-180, 180, (B, N, 1))], -1)
    return ball, rob[:, :nb].copy(), rob[:, nb:].copy()


class _Mirror:
    """One fused handle of the library and one oracle env per env id, driven through the same calls."""

    def __init__(self, L, O, name, seed, rng, log):
        self.L, self.O, self.rng, self.log = L, O, rng, log
        (self.task, self.kind, self.ft, self.nb, self.ny, self.B, self.max_steps, self.layouts) = CONFIGS[name] if isinstance(name, str) else name[:8]
        self.ts = 25 if isinstance(name, str) else name[8]
        self.seed, self.base = seed, int(rng.integers(0, 1 << 20))
        self.layout = None
        self.captured = False
        self.sim = self._handle(self.layouts[int(rng.integers(len(self.layouts)))])
        self.refs = self._oracles(self.seed)
        self.fp = self.sim.get_field_params()
        self.steps_taken = 0

    def _handle(self, layout):
        os.environ["RSX_LAYOUT"] = layout
        try:
            sim = self.L.Sim(self.kind, self.ft, self.nb, self.ny, self.ts, self.B)
            sim.task_attach(self.task, self.seed, self.base, self.max_steps)
        finally:
            del os.environ["RSX_LAYOUT"]
        self.layout = layout
        self.tens = sim.task_tensors()
        return sim

    def _oracles(self, seed):
        refs = [self.O.OracleEnv(self.kind, self.ft, self.nb, self.ny, self.ts, "f32") for _ in range(self.B)]
        for e, r in enumerate(refs):
            r.task_attach(self.task, seed, self.base + e, self.max_steps)
        return refs

    # ---- the calls ----
    def op_reset(self):
        self.sim.task_reset()
        for r in self.refs:
            r.task_reset()

    def op_reset_to(self):
        f = self.fp
        if self.rng.random() < 0.3:
            ball, blue, yellow = wild_placement(self.rng, self.B, self.nb, self.ny, f)
        else:
            ball, blue, yellow = random_placement(self.rng, self.B, self.nb, self.ny, f["length"] / 2 - 3 * f["rbt_radius"],
                                                  f["width"] / 2 - 3 * f["rbt_radius"], 2.3 * f["rbt_radius"], 0.9)
        mask = (self.rng.random(self.B) < 0.4).astype(np.uint8) if self.rng.random() < 0.8 else None
        self.sim.task_reset_to(ball, blue, yellow if self.ny else None, mask)
        for e, r in enumerate(self.refs):
            if mask is None or mask[e]:
                r.task_reset_to(ball[e], blue[e], yellow[e] if self.ny else np.zeros(0))

    def op_step_fed(self):
        import torch
        a = self.rng.uniform(-1, 1, (self.B, self.sim.act_dim)).astype(np.float32)
        self.tens["actions"].copy_(torch.from_numpy(a))
        self.sim.task_step(self.tens["actions"].data_ptr())
        for e, r in enumerate(self.refs):
            r.task_step(a[e])
        self.steps_taken += 1

    def _random_steps(self, n):
        for r in self.refs:
            for _ in range(n):
                r.task_step(None)
        self.steps_taken += n

    def op_step_random(self):
        self.sim.task_step(None)
        self._random_steps(1)

    def op_step_n(self):
        n = int(self.rng.integers(1, 8))
        self.sim.task_step_n(n)
        self._random_steps(n)

    def op_rollout(self):
        n = int(self.rng.integers(1, 8))
        self.sim.task_rollout(n)
        self._random_steps(n)

    def op_checkpoint_to_new_handle(self):
        """save, destroy, create a handle stepped by another layout (never reset), load: the run continues there"""
        blob = self.sim.task_checkpoint()
        was_captured = self.captured
        self.sim.close()
        self.sim = self._handle(self.layouts[int(self.rng.integers(len(self.layouts)))])
        if was_captured and self.rng.random() < 0.5:    # the counter's home (host / device) is a property of the handle, not of the run
            self.sim.task_enable_capture()
        else:
            self.captured = False
        self.sim.task_restore(blob)

    def op_enable_capture(self):
        self.sim.task_enable_capture()
        self.captured = True

    def op_reseed(self):
        self.seed = int(self.rng.integers(0, 1 << 62))
        self.sim.task_reseed(self.seed)
        self.refs = self._oracles(self.seed)
        self.steps_taken = 0
        self.op_reset() if self.rng.random() < 0.5 else self.op_reset_to_all()

    def op_reset_to_all(self):
        f = self.fp
        ball, blue, yellow = random_placement(self.rng, self.B, self.nb, self.ny, f["length"] / 2 - 3 * f["rbt_radius"],
                                              f["width"] / 2 - 3 * f["rbt_radius"], 2.3 * f["rbt_radius"], 0.9)
        self.sim.task_reset_to(ball, blue, yellow if self.ny else None, None)
        for e, r in enumerate(self.refs):
            r.task_reset_to(ball[e], blue[e], yellow[e] if self.ny else np.zeros(0))

    def _teleport(self, cols):
        """overwrite state entries of every env behind the task's back (rsx_set_state on a fused handle: the simulator state only —
        include/rsx.h); mirrored into the oracle.  SSL tasks only: the one-lane VSS-v0 kernel derives its task scalar from the ball it
        finds (documented there), so the layouts agree after such a write for the SSL tasks and are not asked to for VSS-v0"""
        s = self.sim.get_state_full()
        for c, v in cols.items():
            s[:, c] = np.float32(v)
        self.sim.set_state(s)
        for e, r in enumerate(self.refs):
            r.set_state_full(s[e])

    def op_teleport(self):
        if self.task == 1:   # (VSS-v0: also a checkpoint re-derives the previous potential from the ball it saves — rsx.h: rsx_set_state)
            return self.op_step_random()
        f = self.fp
        hx, hy = f["length"] / 2 + 0.2, f["width"] / 2 + 0.2
        what = int(self.rng.integers(0, 3))
        if what == 0:      # the ball anywhere, also beyond the lines and in the goals, at speed
            self._teleport({0: self.rng.uniform(-hx, hx), 1: self.rng.uniform(-hy, hy), 3: self.rng.uniform(-3, 3), 4: self.rng.uniform(-3, 3)})
        elif what == 1:    # the agent robot anywhere (out-of-bounds and distance branches)
            self._teleport({5: self.rng.uniform(-hx, hx), 6: self.rng.uniform(-hy, hy)})
        else:              # the ball stops dead
            self._teleport({3: 0.0, 4: 0.0})

    def op_course(self):
        """SSLDribbling: the ball carried across y = 0 inside the x band of each checkpoint in turn (dribbling.py:155-183), one step after
        each move — the checkpoint counter, its observation entry, the reward and the end after seven on the device, in both layouts"""
        if self.task != 3:
            return self.op_step_random()
        import torch
        bands = [(-0.75, 1), (-1.25, -1), (-1.75, 1), (-2.5, -1), (-1.75, 1), (-2.5, -1), (-1.75, 1)]
        wrong = self.rng.random() < 0.25
        for k, (x, side) in enumerate(bands[: int(self.rng.integers(2, 8))]):
            if wrong and k == 3:
                side = -side
            # 2 cm on one side of the line, moving across it at 2 m/s: the step itself carries the ball over (the reward compares the
            # ball before and after a step); the agent parked away from it, inside its bounds
            self._teleport({0: x, 1: 0.02 * side, 3: 0.0, 4: -2.0 * side, 5: -1.0, 6: 0.5, 8: 0.0, 9: 0.0})
            a = np.zeros((self.B, self.sim.act_dim), dtype=np.float32)
            self.tens["actions"].copy_(torch.from_numpy(a))
            self.sim.task_step(self.tens["actions"].data_ptr())
            for e, r in enumerate(self.refs):
                r.task_step(a[e])
            self.steps_taken += 1
            self.compare("course")

    def op_graph_replays(self):
        """capture one fed-action step into a hipGraph and replay it a few times (only device-keyed handles may)"""
        import torch
        if not self.captured:
            self.op_enable_capture()
        a = self.rng.uniform(-1, 1, (self.B, self.sim.act_dim)).astype(np.float32)
        self.tens["actions"].copy_(torch.from_numpy(a))
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):   # (captures on a side stream of its own)
            self.sim.task_step(self.tens["actions"].data_ptr(), torch.cuda.current_stream().cuda_stream)
        n = int(self.rng.integers(1, 5))
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        for e, r in enumerate(self.refs):
            for _ in range(n):
                r.task_step(a[e])
        self.steps_taken += n

    # ---- the check ----
    def compare(self, what):
        import torch
        torch.cuda.synchronize()
        t = self.tens
        obs, rew = t["obs"].cpu().numpy(), t["reward"].cpu().numpy()
        term, trunc = t["terminated"].cpu().numpy(), t["truncated"].cpu().numpy()
        info, steps, fin = t["info"].cpu().numpy(), t["steps"].cpu().numpy(), t["final_obs"].cpu().numpy()
        state = self.sim.get_state_full()
        ctx = f"{what} [{self.layout}{', device-keyed' if self.captured else ''}] after: " + " ".join(self.log[-12:])
        for e, r in enumerate(self.refs):
            o = r.task_out()
            w = r.get_state_full()
            assert f32_equal(state[e], w), mismatch_report(state[e], w, f"state env {e}: {ctx}")
            assert f32_equal(obs[e], o["obs"]), mismatch_report(obs[e], o["obs"], f"obs env {e}: {ctx}")
            assert steps[e] == o["steps"], f"steps env {e}: {steps[e]} vs {o['steps']}: {ctx}"
            # (a reset writes the state and the observation and clears the step count and the info rows; reward and flags stay those of the
            # env's last step — on both sides — until its next one: reset() returns (obs, {}), vss_gym_base.py:92-106)
            assert f32_equal(rew[e], o["reward"]), f"reward env {e}: {rew[e]} vs {o['reward']}: {ctx}"
            assert term[e] == o["terminated"] and trunc[e] == o["truncated"], f"flags env {e}: {ctx}"
            assert f32_equal(info[:, e], o["info"]), mismatch_report(info[:, e], o["info"], f"info env {e}: {ctx}")
            if o["terminated"] or o["truncated"]:
                assert f32_equal(fin[e], o["final_obs"]), f"final_obs env {e}: {ctx}"

    def compare_metrics(self):
        got = self.sim.read_metrics()
        want = sum(r.task_out()["metrics"] for r in self.refs)
        assert np.array_equal(got, want), (got, want, self.log[-12:])
        assert self.sim.task_tick() == self.steps_taken, (self.sim.task_tick(), self.steps_taken, self.log[-12:])


class _VecMirror(_Mirror):
    """the same sequences through the Python layer a trainer uses (rsoccer_amd.vec: gymnasium.vector-shaped envs over the fused handles):
    actions as numpy arrays, as device tensors of the right and of the wrong dtype / layout / device, step_async + step_wait,
    reset(seed=...), checkpoint() / restore() into a new env, enable_graph_capture()"""
    CLASSES = {1: "VecVSSEnv", 2: "VecSSLStaticDefendersEnv", 3: "VecSSLDribblingEnv", 4: "VecSSLContestedPossessionEnv", 5: "VecSSLPassEnduranceEnv"}

    def _handle(self, layout):
        from rsoccer_amd import vec
        os.environ["RSX_LAYOUT"] = layout
        try:
            if self.task in self.CLASSES:
                cls, kw = getattr(vec, self.CLASSES[self.task]), {}
                if self.task == 1 and self.nb != 3:   # the 5v5 field: the same class with other team sizes (vss/README.md:4; no id is registered for it)
                    cls, kw = type("VecVSS5v5Env", (cls,), dict(N_BLUE=self.nb, N_YELLOW=self.ny)), dict(field_type=self.ft)
                self.env = cls(self.B, seed=self.seed, env_id_base=self.base, max_episode_steps=self.max_steps, **kw)
            else:
                self.env = vec.VecSSLScrimmageEnv(self.B, self.nb, self.ny, self.ft, crowded=self.task == 7, seed=self.seed, env_id_base=self.base,
                                                  max_episode_steps=self.max_steps)
        finally:
            del os.environ["RSX_LAYOUT"]
        self.layout = layout
        self.tens = self.env._t
        return self.env.sim

    def op_reset(self):
        obs, info = self.env.reset()
        assert info == {} and obs.data_ptr() == self.tens["obs"].data_ptr()
        for r in self.refs:
            r.task_reset()

    def _step_any(self, a):
        import torch
        how = int(self.rng.integers(0, 6))
        if how == 0:
            out = self.env.step(a)                                               # numpy float32
        elif how == 1:
            out = self.env.step(a.astype(np.float64))                            # numpy float64 (values are float32-exact)
        elif how == 2:
            out = self.env.step(torch.from_numpy(a).cuda())                      # device tensor, right dtype
        elif how == 3:
            out = self.env.step(torch.from_numpy(a.astype(np.float64)))          # CPU tensor, wrong dtype
        elif how == 4:
            wide = torch.zeros(self.B, 2 * a.shape[1], device="cuda")            # non-contiguous view
            wide[:, ::2] = torch.from_numpy(a).cuda()
            out = self.env.step(wide[:, ::2])
        else:
            self.env.step_async(a)
            out = self.env.step_wait(synchronize=bool(self.rng.random() < 0.5))
        obs, rew, term, trunc, info = out
        assert obs.data_ptr() == self.tens["obs"].data_ptr() and "final_obs" in info and "episode_steps" in info

    def op_step_fed(self):
        a = self.rng.uniform(-1, 1, (self.B, self.sim.act_dim)).astype(np.float32)
        self._step_any(a)
        for e, r in enumerate(self.refs):
            r.task_step(a[e])
        self.steps_taken += 1

    def op_step_random(self):
        self.env.step(None)
        self._random_steps(1)

    def op_step_n(self):
        n = int(self.rng.integers(1, 8))
        self.env.step_random(n)
        self._random_steps(n)

    def op_rollout(self):
        n = int(self.rng.integers(1, 8))
        self.env.step_random(n, fused=True)
        self._random_steps(n)

    def op_reset_to(self):
        f = self.fp
        ball, blue, yellow = random_placement(self.rng, self.B, self.nb, self.ny, f["length"] / 2 - 3 * f["rbt_radius"],
                                              f["width"] / 2 - 3 * f["rbt_radius"], 2.3 * f["rbt_radius"], 0.9)
        mask = (self.rng.random(self.B) < 0.4).astype(np.uint8) if self.rng.random() < 0.8 else None
        self.env.reset_to(ball, blue, yellow if self.ny else None, mask)
        for e, r in enumerate(self.refs):
            if mask is None or mask[e]:
                r.task_reset_to(ball[e], blue[e], yellow[e] if self.ny else np.zeros(0))

    def op_reset_to_all(self):
        self.op_reset()

    def op_checkpoint_to_new_handle(self):
        blob = self.env.checkpoint()
        if self.rng.random() < 0.5:
            blob = blob.tobytes()                    # as it would come back from a file
        was_captured = self.captured
        self.env.close()
        self.sim = self._handle(self.layouts[int(self.rng.integers(len(self.layouts)))])
        if was_captured and self.rng.random() < 0.5:
            self.env.enable_graph_capture()
        else:
            self.captured = False
        self.env.restore(blob)

    def op_enable_capture(self):
        assert self.env.enable_graph_capture() is self.env
        self.captured = True

    def op_reseed(self):
        self.seed = int(self.rng.integers(0, 1 << 62))
        self.env.reset(seed=self.seed)               # gymnasium's way to seed: re-keys every stream, then resets
        self.refs = self._oracles(self.seed)
        for r in self.refs:
            r.task_reset()
        self.steps_taken = 0

    def op_graph_replays(self):
        import torch
        if not self.captured:
            self.op_enable_capture()
        a = self.rng.uniform(-1, 1, (self.B, self.sim.act_dim)).astype(np.float32)
        act = torch.from_numpy(a).cuda()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.env.step(act)
        n = int(self.rng.integers(1, 5))
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        for e, r in enumerate(self.refs):
            for _ in range(n):
                r.task_step(a[e])
        self.steps_taken += n

    def compare_metrics(self):
        m = self.env.metrics()
        want = sum(r.task_out()["metrics"] for r in self.refs)
        assert m["env_steps"] == want[0] and m["episodes"] == want[1], (m, want, self.log[-12:])
        super().compare_metrics()


OPS = [("step_fed", 5), ("step_random", 4), ("step_n", 3), ("rollout", 3), ("reset", 1), ("reset_to", 2),
       ("checkpoint_to_new_handle", 2), ("enable_capture", 1), ("reseed", 1), ("graph_replays", 1), ("metrics", 2), ("teleport", 2), ("course", 1)]


def run_sequence(L, O, name, seed, n_ops, through_vec=False):
    rng = np.random.default_rng(seed)
    log = []
    m = (_VecMirror if through_vec else _Mirror)(L, O, name, int(rng.integers(0, 1 << 62)), rng, log)
    names = [o for o, _ in OPS]
    p = np.array([w for _, w in OPS], dtype=np.float64)
    p /= p.sum()
    try:
        m.op_reset() if rng.random() < 0.5 else m.op_reset_to_all()
        log.append("reset")
        m.compare("reset")
        for _ in range(n_ops):
            op = names[int(rng.choice(len(names), p=p))]
            if op == "enable_capture" and m.captured:
                continue
            log.append(op)
            if op == "metrics":
                m.compare_metrics()
                continue
            getattr(m, "op_" + op)()
            m.compare(op)
        m.compare_metrics()
    finally:
        m.sim.close()
    return log


# ---------------------------------------------------------------------------------------------------------------------------
# the simulator surface without a task: what replaces robosim.{VSS,SSL} (rsim.py:38,102,105,116,155,158,169), batched, in its host
# (float64 wire format) and device (SoA float32) forms — the calls a caller may mix on one handle
# ---------------------------------------------------------------------------------------------------------------------------
RAW_CONFIGS = {
    # kind, field type, n_blue, n_yellow, batch (<= 64: the zero-copy host path; above: the pinned wire buffers)
    "VSS-3v3-small": (0, 0, 3, 3, 5),
    "VSS-3v3": (0, 0, 3, 3, 70),
    "SSL-1v6-small": (1, 2, 1, 6, 3),
    "SSL-1v6": (1, 2, 1, 6, 67),
    "SSL-2v0": (1, 0, 2, 0, 66),
    "SSL-11v11": (1, 1, 11, 11, 9),
}


class _RawMirror:
    def __init__(self, L, O, name, rng, log):
        import torch
        self.torch, self.L, self.rng, self.log = torch, L, rng, log
        self.kind, self.ft, self.nb, self.ny, self.B = RAW_CONFIGS[name] if isinstance(name, str) else name[:5]
        self.ts = 25 if isinstance(name, str) else name[5]
        self.N = self.nb + self.ny
        knobs = {}
        if rng.random() < 0.25:
            knobs["RSX_NO_ZERO_COPY" if self.B <= 64 else "RSX_NO_WIRE_PATH"] = "1"
        os.environ.update(knobs)
        try:
            self.sim = L.Sim(self.kind, self.ft, self.nb, self.ny, self.ts, self.B)
        finally:
            for k in knobs:
                del os.environ[k]
        self.knobs = ",".join(knobs) or "default"
        self.refs = [O.OracleEnv(self.kind, self.ft, self.nb, self.ny, self.ts, "f32") for _ in range(self.B)]
        self.fp = self.sim.get_field_params()
        self.C = self.sim.cmd_dim

    def _cmds(self):
        rng, B, N = self.rng, self.B, self.N
        if self.kind == 0:
            return rng.uniform(-60, 60, (B, N, 2))
        c = np.zeros((B, N, 8))
        wheels = rng.random((B, N)) < 0.3
        c[..., 0] = wheels
        c[..., 1:5] = np.where(wheels[..., None], rng.uniform(-120, 120, (B, N, 4)),
                               np.concatenate([rng.uniform(-3, 3, (B, N, 2)), rng.uniform(-12, 12, (B, N, 1)), np.zeros((B, N, 1))], -1))
        c[..., 5] = np.where(rng.random((B, N)) < 0.3, 4.0, 0.0)
        c[..., 6] = np.where(rng.random((B, N)) < 0.1, 2.0, 0.0)
        c[..., 7] = rng.random((B, N)) < 0.5
        return c

    def _placement(self):
        f = self.fp
        return random_placement(self.rng, self.B, self.nb, self.ny, f["length"] / 2 - 2 * f["rbt_radius"], f["width"] / 2 - 2 * f["rbt_radius"],
                                2.2 * f["rbt_radius"], float(self.rng.choice([0.3, 0.6, 1.0])))

    def _mask(self):
        return (self.rng.random(self.B) < 0.5).astype(np.uint8) if self.rng.random() < 0.6 else None

    def _mirror_reset(self, ball, blue, yellow, mask):
        for e, r in enumerate(self.refs):
            if mask is None or mask[e]:
                r.reset(ball[e], blue[e], yellow[e] if self.ny else np.zeros(0))

    def _mirror_step(self, cmds):
        for e, r in enumerate(self.refs):
            r.step(cmds[e])

    # ---- host-format calls ----
    def op_reset(self):
        ball, blue, yellow = self._placement()
        mask = self._mask()
        self.sim.reset(ball, blue, yellow if self.ny else None, mask)
        self._mirror_reset(ball, blue, yellow, mask)

    def op_reset_wild(self):
        ball, blue, yellow = wild_placement(self.rng, self.B, self.nb, self.ny, self.fp)
        mask = self._mask()
        self.sim.reset(ball, blue, yellow if self.ny else None, mask)
        self._mirror_reset(ball, blue, yellow, mask)

    def op_set_state_wild(self):
        """velocities far beyond what the actuation reaches, the ball in flight"""
        s = self.sim.get_state_full()
        rs = 6 if self.kind == 0 else 11
        for k in range(self.N):
            s[:, 5 + rs * k + 3: 5 + rs * k + 5] = self.rng.uniform(-8, 8, (self.B, 2)).astype(np.float32)
            s[:, 5 + rs * k + 5] = self.rng.uniform(-2000, 2000, self.B).astype(np.float32)
        s[:, 3:5] = self.rng.uniform(-12, 12, (self.B, 2)).astype(np.float32)
        s[:, 2] = (s[:, 2] + np.where(self.rng.random(self.B) < 0.5, self.rng.uniform(0, 0.4, self.B), 0.0)).astype(np.float32)
        s[:, -2] = self.rng.uniform(-3, 3, self.B).astype(np.float32)      # vz
        s[:, -1] = self.rng.uniform(-200, 200, self.B).astype(np.float32)  # spin
        self.sim.set_state(s)
        for e, r in enumerate(self.refs):
            r.set_state_full(s[e])

    def op_step(self):
        c = self._cmds()
        if self.rng.random() < 0.15:   # commands far beyond the motors' range (clamped on both sides; rsim.py:92-101 / :129-153 set no limit)
            if self.kind == 0:
                c = c * 1e6                      # wheel speeds, rad/s
            else:
                c[..., 1:5] = c[..., 1:5] * 1e4  # wheel speeds or local velocities; the flag and the kick / dribbler columns stay what they are
        self.sim.step(c)
        self._mirror_step(c)

    def op_step_state(self):
        c = np.ascontiguousarray(self._cmds())
        got = self.sim.step_state(c, copy=bool(self.rng.random() < 0.5))
        self._mirror_step(c)
        for e, r in enumerate(self.refs):
            assert f32_equal(got[e], r.get_state()), mismatch_report(got[e], r.get_state(), f"step_state env {e}: " + " ".join(self.log[-10:]))

    def op_step_wire(self):
        w = self.sim.wire_buffers()
        if w is None:
            return self.op_step()
        c = self._cmds()
        np.copyto(w[0], c)
        self.sim.step_wire()
        self._mirror_step(c)
        for e, r in enumerate(self.refs):
            assert f32_equal(w[1][e], r.get_state_full()), mismatch_report(w[1][e], r.get_state_full(), f"wire state env {e}: " + " ".join(self.log[-10:]))

    def op_set_state(self):
        s = self.sim.get_state_full()
        k = self.rng.integers(0, s.shape[1], 6)
        if self.kind == 1:   # the infrared column of an SSL robot is a flag (Frame.py:86 reads it with bool()): 0 or 1 stays 0 or 1
            k = np.array([c for c in k if not (c >= 5 and c < 5 + 11 * self.N and (c - 5) % 11 == 6)], dtype=int)
        s[:, k] = (s[:, k] + self.rng.normal(0, 0.05, (self.B, len(k)))).astype(np.float32)
        self.sim.set_state(s)
        for e, r in enumerate(self.refs):
            r.set_state_full(s[e])

    # ---- device-resident calls ----
    def op_step_dev(self, flip=False):
        torch = self.torch
        c = self._cmds().astype(np.float32)
        self.sim.cmds_tensor().copy_(torch.from_numpy(np.ascontiguousarray(c.reshape(self.B, -1).T)))
        (self.sim.step_dev_flip if flip else self.sim.step_dev)()
        self._mirror_step(c)

    def op_step_dev_flip(self):
        self.op_step_dev(flip=True)
        cur, oth = self.sim.state_buffers()
        torch = self.torch
        torch.cuda.synchronize()
        got = cur.cpu().numpy().T.astype(np.float64)
        for e, r in enumerate(self.refs):
            assert f32_equal(got[e], r.get_state_full()), mismatch_report(got[e], r.get_state_full(), f"current buffer env {e}: " + " ".join(self.log[-10:]))

    def op_step_dev_random(self):
        n, seed, t0 = int(self.rng.integers(1, 6)), int(self.rng.integers(0, 1 << 40)), int(self.rng.integers(0, 1000))
        self.sim.step_dev_random(n, seed, t0)
        for e, r in enumerate(self.refs):
            for t in range(n):
                r.step_random(seed, e, t0 + t)

    def op_reset_dev(self):
        torch = self.torch
        ball, blue, yellow = (a.astype(np.float32) for a in self._placement())
        mask = self._mask()
        d = lambda a: torch.from_numpy(a).cuda()
        self.sim.reset_dev(d(ball), d(blue) if self.nb else None, d(yellow) if self.ny else None, None if mask is None else d(mask))
        self._mirror_reset(ball, blue, yellow, mask)

    def op_device_write(self):
        """a caller's own kernel writes the state array (here: torch): the next host-format read must see it"""
        torch = self.torch
        row = int(self.rng.integers(0, 5))
        v = self.rng.uniform(-0.3, 0.3, self.B).astype(np.float32)
        cur = self.sim.state_buffers()[0]   # (after rsx_step_dev_flip calls the buffer that holds the current frame)
        cur[row].copy_(torch.from_numpy(v))
        for e, r in enumerate(self.refs):
            s = r.get_state_full()
            s[row] = v[e]
            r.set_state_full(s)

    def compare(self, what):
        self.torch.cuda.synchronize()
        ctx = f"{what} [{self.knobs}] after: " + " ".join(self.log[-12:])
        if self.rng.random() < 0.3:
            got = self.sim.get_state()
            for e, r in enumerate(self.refs):
                assert f32_equal(got[e], r.get_state()), mismatch_report(got[e], r.get_state(), f"get_state env {e}: {ctx}")
        got = self.sim.get_state_full()
        for e, r in enumerate(self.refs):
            w = r.get_state_full()
            assert f32_equal(got[e], w), mismatch_report(got[e], w, f"state env {e}: {ctx}")


RAW_OPS = [("reset", 2), ("reset_wild", 2), ("set_state_wild", 1), ("step", 3), ("step_state", 3), ("step_wire", 2), ("set_state", 1), ("step_dev", 3), ("step_dev_flip", 3),
           ("step_dev_random", 2), ("reset_dev", 2), ("device_write", 1)]


def run_raw_sequence(L, O, name, seed, n_ops):
    rng = np.random.default_rng(seed)
    log = []
    m = _RawMirror(L, O, name, rng, log)
    names = [o for o, _ in RAW_OPS]
    p = np.array([w for _, w in RAW_OPS], dtype=np.float64)
    p /= p.sum()
    try:
        m.compare("create")   # the adapter's dummy line-up, rsim.py:20-24
        for _ in range(n_ops):
            op = names[int(rng.choice(len(names), p=p))]
            log.append(op)
            getattr(m, "op_" + op)()
            m.compare(op)
    finally:
        m.sim.close()
    return log


@pytest.mark.parametrize("name", list(RAW_CONFIGS))
def test_random_call_sequences_of_the_simulator_surface_match_the_oracle(oracle_mod, name):
    from rsoccer_amd import _lib as L
    for seed in (21, 22):
        run_raw_sequence(L, oracle_mod, name, seed * 1000 + sorted(RAW_CONFIGS).index(name), 40)


@pytest.mark.parametrize("seed", [1000068, 1000415])
def test_masked_reset_to_leaves_the_other_envs_observations_alone(oracle_mod, seed):
    """found by the long campaign (3 of 500 SSLDribbling sequences): the refresh launch of rsx_task_reset_to rewrote the observation of
    EVERY env from the current task scalar — for an env the mask left alone whose last step had just passed a checkpoint that is the
    count AFTER the step's reward moved it, not the one the step's observation saw (dribbling.py: observation before reward)"""
    from rsoccer_amd import _lib as L
    run_sequence(L, oracle_mod, "SSLDribbling-v0", seed, 150)


@pytest.mark.parametrize("layout", ["lanes", "epl"])
def test_dribbling_course_earns_its_checkpoints_on_the_device(oracle_mod, layout):
    """the course op of the fuzzer really walks SSLDribbling's checkpoint logic (dribbling.py:155-183) on the device: counters move, the
    seventh checkpoint ends the episode as a success — compared with the oracle after every step, in both kernel layouts"""
    from rsoccer_amd import _lib as L
    CONFIGS["_course"] = (3, 1, 2, 1, 4, 37, 400, (layout,))
    try:
        m = _Mirror(L, oracle_mod, "_course", 5, np.random.default_rng(3), [])
        m.op_reset()
        m.compare("reset")
        for _ in range(8):
            m.op_course()
        got = m.sim.read_metrics()
        assert got[1] > 0 and got[2] > 0, got          # episodes ended, some of them by completing the course
        m.compare_metrics()
        m.sim.close()
    finally:
        del CONFIGS["_course"]


def test_masked_reset_to_leaves_the_other_envs_flags_alone(oracle_mod):
    """the defect this file found: rsx_task_reset_to carried its env mask through the `truncated` row and cleared the row afterwards —
    for every env, also those the mask left alone"""
    import torch
    from rsoccer_amd import _lib as L
    B = 24
    sim = L.Sim(1, 2, 2, 0, 25, B)
    sim.task_attach(5, 1, 0, 3)            # pass endurance, TimeLimit 3: the third step truncates every env
    sim.task_reset()
    sim.task_step_n(3)
    t = sim.task_tensors()
    torch.cuda.synchronize()
    trunc = t["truncated"].cpu().numpy().copy()
    assert trunc.sum() > B // 2
    term = t["terminated"].cpu().numpy().copy()
    rng = np.random.default_rng(0)
    f = sim.get_field_params()
    ball, blue, _ = random_placement(rng, B, 2, 0, f["length"] / 2 - 0.5, f["width"] / 2 - 0.5, 0.3)
    mask = (np.arange(B) % 3 == 0).astype(np.uint8)
    sim.task_reset_to(ball, blue, None, mask)
    torch.cuda.synchronize()
    assert np.array_equal(t["truncated"].cpu().numpy(), trunc) and np.array_equal(t["terminated"].cpu().numpy(), term)
    sim.close()


@pytest.mark.parametrize("name", list(CONFIGS))
def test_random_call_sequences_match_the_oracle(oracle_mod, name):
    from rsoccer_amd import _lib as L
    for seed in (11, 12):
        run_sequence(L, oracle_mod, name, seed * 1000 + sorted(CONFIGS).index(name), 45)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_random_call_sequences_through_the_vector_envs_match_the_oracle(oracle_mod, name):
    from rsoccer_amd import _lib as L
    for seed in (31, 32):
        run_sequence(L, oracle_mod, name, seed * 1000 + sorted(CONFIGS).index(name), 45, through_vec=True)


def random_task_config(rng):
    """team sizes, fields, time steps and TimeLimits no registered id uses: the run-time-robot-count kernel variants of every lane-group width"""
    ts = int(rng.choice([0, 5, 7, 16, 25, 33, 50]))
    B, ms = int(rng.choice([1, 7, 33, 65])), int(rng.integers(1, 30))
    what = int(rng.integers(0, 3))
    if what == 0:      # VSS-v0 arithmetic on any VSS line-up whose observation fits (4 + 7 nb + 5 ny <= 64)
        while True:
            nb, ny = int(rng.integers(1, 6)), int(rng.integers(0, 6))
            if 4 + 7 * nb + 5 * ny <= 64:
                break
        return (1, 0, int(rng.integers(0, 2)), nb, ny, B, ms, ("lanes",), ts)
    if what == 1:      # static defenders 1 v N
        return (2, 1, int(rng.integers(0, 3)), 1, int(rng.integers(0, 11)), B, ms, ("lanes",), ts)
    nb, ny = int(rng.integers(0, 12)), int(rng.integers(0, 12))
    if nb + ny == 0:
        nb = 1
    return (int(rng.choice([6, 7])), 1, int(rng.integers(0, 3)), nb, ny, B, ms, ("lanes",), ts)


def random_raw_config(rng):
    kind = int(rng.integers(0, 2))
    hi = 6 if kind == 0 else 12
    nb, ny = int(rng.integers(0, hi)), int(rng.integers(0, hi))
    if nb + ny == 0:
        ny = 1
    return (kind, int(rng.integers(0, 2 if kind == 0 else 3)), nb, ny, int(rng.choice([1, 9, 64, 65, 100])), int(rng.choice([0, 5, 7, 16, 25, 33, 50])))


if __name__ == "__main__":
    import argparse
    import time
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=20)
    ap.add_argument("--ops", type=int, default=120)
    ap.add_argument("--first-seed", type=int, default=100000)
    ap.add_argument("--configs", default=",".join(CONFIGS))
    ap.add_argument("--random-configs", type=int, default=0, help="that many random team sizes / fields / time steps instead of the named configurations")
    a = ap.parse_args()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    g.build()
    from oracle import oracle as O
    from rsoccer_amd import _lib as L
    O.build()
    bad = 0
    if a.random_configs:
        crng = np.random.default_rng(a.first_seed)
        t0, n_calls = time.time(), 0
        for i in range(a.random_configs):
            raw, cfg = random_raw_config(crng), random_task_config(crng)
            for what, c, run in (("raw", raw, run_raw_sequence), ("task", cfg, run_sequence)):
                try:
                    n_calls += len(run(L, O, c, a.first_seed + i, a.ops))
                except AssertionError as ex:
                    bad += 1
                    print(f"FAIL {what} {c} seed {a.first_seed + i}: {str(ex)[:1500]}", flush=True)
                except L.RsxError as ex:
                    print(f"refused {what} {c}: {ex}", flush=True)
        print(f"random configurations: {a.random_configs} x (simulator surface + fused task), {n_calls} calls compared call by call, {time.time() - t0:.0f} s, failures {bad}", flush=True)
        sys.exit(1 if bad else 0)
    for name in RAW_CONFIGS if a.configs == ",".join(CONFIGS) else []:
        t0 = time.time()
        n_calls = 0
        for sd in range(a.first_seed, a.first_seed + a.seeds):
            try:
                n_calls += len(run_raw_sequence(L, O, name, sd, a.ops))
            except AssertionError as ex:
                bad += 1
                print(f"FAIL raw {name} seed {sd}: {str(ex)[:1500]}", flush=True)
        print(f"simulator surface {name}: {a.seeds} sequences, {n_calls} calls compared call by call, {time.time() - t0:.0f} s, failures so far {bad}", flush=True)
    for name in a.configs.split(","):
        t0 = time.time()
        n_calls = 0
        for s in range(a.first_seed, a.first_seed + a.seeds):
            try:
                n_calls += len(run_sequence(L, O, name, s, a.ops, through_vec=bool(s & 1)))   # odd seeds: through rsoccer_amd.vec
            except AssertionError as ex:
                bad += 1
                print(f"FAIL {name} seed {s}: {str(ex)[:1500]}", flush=True)
        print(f"{name}: {a.seeds} sequences, {n_calls} calls compared call by call, {time.time() - t0:.0f} s, failures so far {bad}", flush=True)
    sys.exit(1 if bad else 0)
