"""Per-env adapter: UNMODIFIED scalar-hook task classes over one batched simulator.

The reference's contract is four hooks over ONE frame (rsoccer_gym/vss/vss_gym_base.py:197-211):
``_get_commands(action)``, ``_frame_to_observations()``, ``_calculate_reward_and_done()``,
``_get_initial_positions_frame()``.  A task written that way (the classes of
``rsoccer_amd.vss`` / ``rsoccer_amd.ssl``, or a user's own subclass of ``VSSBaseEnv`` /
``SSLBaseEnv``) runs here as ``num_envs`` Python objects whose ``rsim`` is a slot of ONE batched
HIP simulator: per ``step()`` the hooks run in a Python loop (that is the slow part, by design —
it is the compatibility path for code the engine has never seen), the physics of all envs is one
``rsx_step`` and one state read-back.  Same-step auto-reset and TimeLimit as ``gymnasium.vector``.
"""
import numpy as np

from rsoccer_amd import _lib
from rsoccer_amd.Entities import Field


class _Slot:
    """what ``RSim*`` expects of ``robosim.VSS`` / ``robosim.SSL`` (rsim.py:38,50,102,105), for env i
    of a shared batch: commands and placements are collected, states are served from the batch."""

    def __init__(self, pool, i):
        self.pool, self.i = pool, i

    def step(self, cmds):
        self.pool.cmds[self.i] = cmds

    def get_state(self):
        return self.pool.state[self.i]

    def reset(self, ball, blue, yellow):
        self.pool.place(self.i, ball, blue, yellow)

    def get_field_params(self):
        return self.pool.field

    def close(self):
        pass


class _Backend:
    """stands where the ``robosim`` module stands (``sim_backend=`` of the base envs); hands out the
    slots of the pool in construction order"""

    def __init__(self, pool):
        self.pool, self.n = pool, 0

    def _make(self, field_type, n_blue, n_yellow, time_step_ms, ball, blue, yellow):
        pool = self.pool
        pool.bind(field_type, n_blue, n_yellow, time_step_ms)
        slot = _Slot(pool, self.n)
        self.n += 1
        return slot

    def VSS(self, *a):
        self.pool.kind = _lib.KIND_VSS
        return self._make(*a)

    def SSL(self, *a):
        self.pool.kind = _lib.KIND_SSL
        return self._make(*a)


class _Pool:
    def __init__(self, num_envs, device):
        self.B, self.device = num_envs, device
        self.sim = None
        self.kind = None
        self.pending = np.zeros(num_envs, dtype=np.uint8)

    def bind(self, field_type, n_blue, n_yellow, time_step_ms):
        if self.sim is not None:
            return
        self.sim = _lib.Sim(self.kind, field_type, n_blue, n_yellow, time_step_ms, self.B, self.device)
        self.nb, self.ny = n_blue, n_yellow
        self.field = self.sim.get_field_params()
        self.cmds = np.zeros((self.B, n_blue + n_yellow, self.sim.cmd_dim))
        self.state = self.sim.get_state()
        self.ball = np.zeros((self.B, 4))
        self.blue = np.zeros((self.B, n_blue, 3))
        self.yellow = np.zeros((self.B, n_yellow, 3))

    def place(self, i, ball, blue, yellow):
        """robosim.reset of env i: recorded (flushed as one masked batch reset before the next step);
        the state it will have is known without asking the device (rsim.py:52-75: poses, zero rates)"""
        self.ball[i] = ball
        if self.nb:
            self.blue[i] = np.asarray(blue, float).reshape(self.nb, 3)
        if self.ny:
            self.yellow[i] = np.asarray(yellow, float).reshape(self.ny, 3)
        self.pending[i] = 1
        rs = 6 if self.kind == _lib.KIND_VSS else 11
        s = np.zeros(self.sim.state_dim)
        s[0:2] = self.ball[i, 0:2]; s[2] = self.field["ball_radius"]; s[3:5] = self.ball[i, 2:4]
        for k in range(self.nb + self.ny):
            s[5 + rs * k: 8 + rs * k] = self.blue[i, k] if k < self.nb else self.yellow[i, k - self.nb]
        self.state[i] = s.astype(np.float32)   # what the engine will hold (f32)

    def flush(self):
        if self.pending.any():
            self.sim.reset(self.ball, self.blue, self.yellow, self.pending)
            self.pending[:] = 0

    def step(self):
        self.flush()
        self.sim.step(self.cmds)
        self.state = self.sim.get_state()


class VecScalarHookEnv:
    """``VecScalarHookEnv(VSSEnv, 64)``: 64 instances of an unmodified scalar-hook task class on one
    batched simulator.  ``reset()`` -> (obs [B, D], {}); ``step(actions [B, A])`` -> (obs, reward [B],
    terminated [B], truncated [B], info) with same-step auto-reset (``info["final_obs"]``,
    ``info["infos"]`` = the per-env dicts some tasks return)."""

    def __init__(self, env_cls, num_envs, max_episode_steps=None, device=0, **env_kwargs):
        self.num_envs = int(num_envs)
        self.max_episode_steps = max_episode_steps
        self.pool = _Pool(self.num_envs, int(device))
        backend = _Backend(self.pool)
        self.envs = [env_cls(sim_backend=backend, **env_kwargs) for _ in range(self.num_envs)]
        self.single_observation_space = self.envs[0].observation_space
        self.single_action_space = self.envs[0].action_space
        from rsoccer_amd.vec.fused import batched_space
        self.observation_space = batched_space(self.single_observation_space, self.num_envs)
        self.action_space = batched_space(self.single_action_space, self.num_envs)
        self.field = Field(**self.pool.field)
        self.elapsed = np.zeros(self.num_envs, dtype=np.int64)

    def reset(self, *, seed=None, options=None):
        obs = [e.reset(seed=None if seed is None else seed + i)[0] for i, e in enumerate(self.envs)]
        self.pool.flush()
        self.elapsed[:] = 0
        return np.stack(obs), {}

    def step(self, actions):
        actions = np.asarray(actions)
        if actions.shape[0] != self.num_envs:
            raise ValueError(f"actions must have {self.num_envs} rows")
        # the reference's step() template (vss_gym_base.py:72-90), split around ONE batched physics call
        for e, a in zip(self.envs, actions):
            e.steps += 1
            cmds = e._get_commands(a)
            e.rsim.send_commands(cmds)          # -> slot: collected
            e.sent_commands = cmds
        self.pool.step()
        obs, rew, term, infos = [], [], [], []
        for e in self.envs:
            e.last_frame, e.frame = e.frame, e.rsim.get_frame()
            obs.append(e._frame_to_observations())
            out = e._calculate_reward_and_done()
            rew.append(out[0]); term.append(bool(out[1]))
            infos.append(getattr(e, "reward_shaping_total", None) or getattr(e, "reward_info", None) or {})
        self.elapsed += 1
        term = np.asarray(term)
        trunc = self.elapsed >= self.max_episode_steps if self.max_episode_steps else np.zeros(self.num_envs, bool)
        obs = np.stack(obs)
        info = {"final_obs": obs.copy(), "infos": infos}
        for i in np.nonzero(term | trunc)[0]:
            obs[i] = self.envs[i].reset()[0]
            self.elapsed[i] = 0
        return obs, np.asarray(rew, dtype=np.float64), term, trunc, info

    def close(self):
        for e in self.envs:
            e.rsim.simulator = None
        if self.pool.sim is not None:
            self.pool.sim.close()
