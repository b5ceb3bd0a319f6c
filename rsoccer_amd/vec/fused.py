"""Fused batched tasks (one kernel launch per ``step()``)."""
import numpy as np

from rsoccer_amd import _lib
from rsoccer_amd import gymshim as gym

_VSS_INFO = ("goal_score", "move", "ball_grad", "energy", "goals_blue", "goals_yellow")
_SD_INFO = ("goal", "rbt_in_gk_area", "done_ball_out", "done_ball_out_right", "done_rbt_out",
            "ball_dist", "ball_grad", "energy")


def batched_space(single, n):
    """the space of ``n`` copies of ``single`` (a Box), as ``gymnasium.vector`` describes a vector env: shape ``(n,) + single.shape``"""
    low = np.broadcast_to(single.low, (n,) + tuple(single.shape)).copy()
    high = np.broadcast_to(single.high, (n,) + tuple(single.shape)).copy()
    return gym.spaces.Box(low=low, high=high, shape=(n,) + tuple(single.shape), dtype=single.dtype)


class VecFusedEnv:
    """``num_envs`` copies of a fused task on one GPU.

    ``reset()`` -> ``(obs, info)``; ``step(actions)`` -> ``(obs, reward, terminated, truncated,
    info)`` like ``gymnasium.vector`` with same-step auto-reset: when an env's episode ends
    (task ``done`` or the registry's TimeLimit) it is re-placed inside the same call, ``obs``
    already holds the first observation of its next episode and ``info["final_obs"]`` the
    terminal one.  All returned arrays are torch tensors on the env's device that VIEW the
    engine's buffers (no copies): they are overwritten by the next ``step()``.

    Every random draw (placement, OU noise, random actions) is keyed by
    ``(seed, env_id_base + env index, episode, step)``, so results do not depend on
    ``num_envs`` or on how a population of envs is split over GPUs.
    """

    KIND = None
    TASK = None
    FIELD_TYPE = 0
    N_BLUE = N_YELLOW = 0
    INFO_KEYS = ()
    TIME_STEP = 0.025

    def __init__(self, num_envs, device=0, seed=0, env_id_base=0, max_episode_steps=None,
                 field_type=None):
        import torch
        self._torch = torch
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", int(device))
        ft = self.FIELD_TYPE if field_type is None else field_type
        self.sim = _lib.Sim(self.KIND, ft, self.N_BLUE, self.N_YELLOW, int(self.TIME_STEP * 1000),
                            self.num_envs, int(device))
        self.sim.task_attach(self.TASK, int(seed), int(env_id_base), int(max_episode_steps or 0))
        self.max_episode_steps = self.sim.max_episode_steps
        self._t = self.sim.task_tensors()
        self._info_views = None
        self._pending = None
        self.field = self.sim.get_field_params()
        self.single_action_space = gym.spaces.Box(low=-1, high=1, shape=(self.sim.act_dim,), dtype=np.float32)
        self.single_observation_space = gym.spaces.Box(low=-1.2, high=1.2, shape=(self.sim.obs_dim,), dtype=np.float32)
        # gymnasium.vector convention: action_space / observation_space describe the batch, single_* one env
        self.action_space = batched_space(self.single_action_space, self.num_envs)
        self.observation_space = batched_space(self.single_observation_space, self.num_envs)

    # ---- gym-like surface ----
    def _stream(self):
        # raw handle of torch's current stream on the env's device (the private getter skips the
        # Stream object; the public API is the fallback)
        try:
            return self._torch._C._cuda_getCurrentRawStream(self.device.index)
        except AttributeError:
            return self._torch.cuda.current_stream(self.device).cuda_stream

    def _info(self):
        """The info dict of step(): tensor VIEWS of the engine's buffers, so one dict serves every
        step (a fresh shallow copy is returned: callers may add keys)."""
        if self._info_views is None:
            t = self._t
            info = {k: t["info"][i] for i, k in enumerate(self.INFO_KEYS)}
            info["final_obs"] = t["final_obs"]
            info["episode_steps"] = t["steps"]
            self._info_views = info
        return dict(self._info_views)

    def reset(self, *, seed=None, options=None):
        """New random placement for every env.  ``seed=None`` (the usual call): the random streams chosen at construction go on
        (episode counters advance).  ``seed=<int>``: start over — every random stream (placements, OU noise, random actions) is
        re-keyed and all counters cleared, exactly as if the env had just been constructed with that seed
        (``rsx_task_reseed``; env ``i`` keeps its global id ``env_id_base + i``, which distinguishes the envs' streams)."""
        if seed is not None:
            self.sim.task_reseed(int(seed), self._stream())
        self.sim.task_reset(self._stream())
        return self._t["obs"], {}

    def reset_to(self, ball, blue, yellow, env_mask=None):
        """Start new episodes on explicit placements (arrays as ``robosim.reset``: ball [B,4],
        blue [B,nb,3], yellow [B,ny,3]); ``env_mask`` selects the envs to touch."""
        self.sim.task_reset_to(ball, blue, yellow, env_mask, self._stream())
        return self._t["obs"], {}

    def step(self, actions=None):
        """actions: ``[num_envs, act_dim]`` float32 (torch CUDA tensor = zero-copy; numpy is
        staged through a device buffer) or None for uniform random actions drawn on device."""
        torch = self._torch
        ptr = None
        if actions is not None:
            want = (self.num_envs, self.sim.act_dim)
            if tuple(np.shape(actions)) != want:   # checked on the INPUT: copy_ below would broadcast
                raise ValueError(f"actions must be [{want[0]}, {want[1]}], got {tuple(np.shape(actions))}")
            if isinstance(actions, torch.Tensor):
                a = actions
                if a.device != self.device or a.dtype != torch.float32 or not a.is_contiguous():
                    a = a.to(device=self.device, dtype=torch.float32).contiguous()
            else:
                a = self._t["actions"]
                # pageable host memory: a blocking copy (non_blocking would read a temporary)
                a.copy_(torch.from_numpy(np.ascontiguousarray(actions, dtype=np.float32)))
            self._keep = a  # keep the tensor alive until the launch has consumed it
            ptr = a.data_ptr()
        self.sim.task_step(ptr, self._stream())
        t = self._t
        return t["obs"], t["reward"], t["terminated"], t["truncated"], self._info()

    def enable_graph_capture(self):
        """Make ``step()`` / ``step_random()`` capturable into a ``torch.cuda.CUDAGraph`` (hipGraph) and replayable.

        The engine's step counter — the key of the per-step random draws — moves to device memory
        (``rsx_task_enable_capture``), so every replay advances it exactly as an eager call would: a run is
        bit-identical whether its steps are issued eagerly, replayed from a graph, or both.  Call once, outside any
        capture; without it a captured ``step()`` raises instead of silently replaying one random stream.  Typical use
        (the loop of the reference's README.md:116-133 with the policy on the GPU)::

            env.enable_graph_capture()
            actions = torch.zeros(env.num_envs, env.sim.act_dim, device=env.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                actions.copy_(policy(obs))
                env.step(actions)
            for _ in range(n): g.replay()      # obs / reward / flags are the same tensor views as ever
        """
        self.sim.task_enable_capture(self._stream())
        return self

    def step_async(self, actions=None):
        """``gymnasium.vector``-style split call: enqueue the step (host-asynchronous, stream-ordered,
        exactly what ``step`` does)."""
        self._pending = self.step(actions)

    def step_wait(self, synchronize=False):
        """Results of the last ``step_async``: device tensors that are valid in stream order;
        ``synchronize=True`` also blocks the host until the launch has finished."""
        out, self._pending = self._pending, None
        if out is None:
            raise RuntimeError("step_wait() without step_async()")
        if synchronize:
            self._torch.cuda.current_stream(self.device).synchronize()
        return out

    def step_random(self, n=1, fused=False):
        """``n`` steps with device-side random actions: ``n`` launches issued from C, or
        (``fused=True``) one launch that keeps the state in registers between steps."""
        (self.sim.task_rollout if fused else self.sim.task_step_n)(int(n), self._stream())
        t = self._t
        return t["obs"], t["reward"], t["terminated"], t["truncated"], self._info()

    def checkpoint(self):
        """Everything needed to continue this run bit-identically (numpy uint8 blob, ``rsx_task_checkpoint_save``):
        simulator state, episode bookkeeping, noise state, the random streams' step counter, metrics.  Restore
        with ``restore()`` on an env of the same class, size, seed and ``env_id_base`` — in another process or on
        another GPU."""
        return self.sim.task_checkpoint(self._stream())

    def restore(self, blob):
        self.sim.task_restore(blob, self._stream())
        t = self._t
        return t["obs"], self._info()

    def metrics(self):
        """Counters accumulated on device since construction (synchronises the stream)."""
        m = self.sim.read_metrics(self._stream())
        out = dict(zip(_lib.METRIC_NAMES, (int(v) for v in m)))
        out["return_sum"] = out.pop("return_sum_q20") / float(1 << 20)
        return out

    @property
    def state(self):
        """[state_dim + 2, num_envs] float32 view of the SoA simulator state — for READING (frames for logging, rendering, analysis).
        Writing it moves bodies behind the task's back: observations and per-episode task scalars are not refreshed (include/rsx.h:
        rsx_set_state); to re-place envs use ``reset_to``."""
        return self.sim.state_tensor()

    def close(self):
        self.sim.close()


_CONT_INFO = _SD_INFO + ("collision",)


class VecVSSEnv(VecFusedEnv):
    """VSS-v0 (rsoccer_gym/vss/env_vss/vss_gym.py:13): obs 40, action 2, TimeLimit 1200."""
    KIND, TASK, FIELD_TYPE, N_BLUE, N_YELLOW = _lib.KIND_VSS, _lib.TASK_VSS_V0, 0, 3, 3
    INFO_KEYS = _VSS_INFO


class VecSSLStaticDefendersEnv(VecFusedEnv):
    """SSLStaticDefenders-v0 (ssl/ssl_hw_challenge/static_defenders.py:12): obs 24, action 5,
    TimeLimit 1000, hardware-challenge field (field_type 2)."""
    KIND, TASK, FIELD_TYPE, N_BLUE, N_YELLOW = _lib.KIND_SSL, _lib.TASK_SSL_STATIC_DEFENDERS, 2, 1, 6
    INFO_KEYS = _SD_INFO


class VecSSLDribblingEnv(VecFusedEnv):
    """SSLDribbling-v0 (ssl/ssl_hw_challenge/dribbling.py:11): obs 21, action 4, TimeLimit 4800.
    info: the checkpoint counter (the reference task returns an empty info dict)."""
    KIND, TASK, FIELD_TYPE, N_BLUE, N_YELLOW = _lib.KIND_SSL, _lib.TASK_SSL_DRIBBLING, 2, 1, 4
    INFO_KEYS = ("checkpoints",)


class VecSSLContestedPossessionEnv(VecFusedEnv):
    """SSLContestedPossession-v0 (ssl/ssl_hw_challenge/contested_possession.py:11): obs 14,
    action 5, TimeLimit 1200."""
    KIND, TASK, FIELD_TYPE, N_BLUE, N_YELLOW = _lib.KIND_SSL, _lib.TASK_SSL_CONTESTED, 2, 1, 1
    INFO_KEYS = _CONT_INFO


class VecSSLPassEnduranceEnv(VecFusedEnv):
    """SSLPassEndurance-v0 (ssl/ssl_hw_challenge/pass_endurance.py:11): obs 16, action 3,
    TimeLimit 1200."""
    KIND, TASK, FIELD_TYPE, N_BLUE, N_YELLOW = _lib.KIND_SSL, _lib.TASK_SSL_PASS_ENDURANCE, 2, 2, 0
    INFO_KEYS = ("reversed_dist", "ball_grad")


class VecSSLScrimmageEnv(VecFusedEnv):
    """Synthetic SSL task in the style of the reference's example env (README.md:78-110) for team sizes
    no registered id covers — BASELINE.json configs[3]: 11v11 on the division-A field.  EVERY robot
    is commanded: action ``[B, 4 N]`` = per robot (v_x, v_y robot-local x 2.5 m/s, v_theta x 10 rad/s,
    kick 5 m/s when > 0.9); obs ``[B, 2 + 2 N]`` = normalised ball and robot positions; reward +1 / -1
    and done on a goal for blue / yellow; TimeLimit 1200.  ``crowded=True`` packs the line-up around
    the ball (worst-case all-pairs contacts)."""
    KIND = _lib.KIND_SSL
    INFO_KEYS = ("goals_blue", "goals_yellow")

    def __init__(self, num_envs, n_blue=11, n_yellow=11, field_type=1, crowded=False, **kw):
        self.N_BLUE, self.N_YELLOW, self.FIELD_TYPE = int(n_blue), int(n_yellow), int(field_type)
        self.TASK = _lib.TASK_SSL_SCRIMMAGE_CROWDED if crowded else _lib.TASK_SSL_SCRIMMAGE
        super().__init__(num_envs, **kw)
