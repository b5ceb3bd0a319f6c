"""Batched form of the reference's subclass contract.

The reference asks a task author for four hooks over ONE frame
(rsoccer_gym/vss/vss_gym_base.py:197-211, ssl/ssl_gym_base.py:197-211).  Here the same four
hooks see ``num_envs`` frames at once: every ``frame.ball.x`` / ``frame.robots_blue[i].theta``
is a ``[B]`` float32 tensor that views one row of the simulator's SoA state (no copy), commands
are written into ``self.commands`` (a ``[n_robots, C, B]`` view of the command buffer), and the
hooks are ordinary torch code running on the device.  The physics in between is one launch of
the raw step kernel (``rsx_step_dev``).
"""
import numpy as np

from rsoccer_amd import _lib
from rsoccer_amd.Entities import Field

_VSS_BLOCK = ("x", "y", "theta", "v_x", "v_y", "v_theta")
_SSL_BLOCK = _VSS_BLOCK + ("infrared", "v_wheel0", "v_wheel1", "v_wheel2", "v_wheel3")


class _Rows:
    """attribute -> row of a [rows, B] tensor"""

    def __init__(self, state, base, names):
        object.__setattr__(self, "_state", state)
        object.__setattr__(self, "_index", {n: base + i for i, n in enumerate(names)})

    def __getattr__(self, name):
        try:
            return self._state[self._index[name]]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self._state[self._index[name]].copy_(value) if hasattr(value, "shape") else self._state[self._index[name]].fill_(value)


class VecFrame:
    """``frame.ball.{x,y,z,v_x,v_y}``, ``frame.robots_blue[i].<field>``, ``frame.robots_yellow[i].<field>``
    as [B] tensors (units of Entities/Frame.py:8: m, m/s, degrees, degrees/s)."""

    def __init__(self, state, n_blue, n_yellow, kind):
        block = _VSS_BLOCK if kind == _lib.KIND_VSS else _SSL_BLOCK
        w = len(block)
        self.state = state
        self.ball = _Rows(state, 0, ("x", "y", "z", "v_x", "v_y"))
        self.robots_blue = {i: _Rows(state, 5 + w * i, block) for i in range(n_blue)}
        self.robots_yellow = {i: _Rows(state, 5 + w * (n_blue + i), block) for i in range(n_yellow)}
        self._shape = (n_blue, n_yellow, kind)

    def clone(self):
        return VecFrame(self.state.clone(), *self._shape)


class _VecBaseEnv:
    """``num_envs`` envs behind the four hooks of the reference's base classes, with the episode
    bookkeeping of ``gymnasium.vector`` done ON THE DEVICE: ``step()`` sets ``truncated`` from the
    registry's TimeLimit (``max_episode_steps``), re-places the envs whose episode ended inside the
    same call (``info["final_obs"]`` keeps their terminal observation) and never copies anything
    between host and device — provided ``_get_initial_positions`` returns device tensors.  The
    previous frame (``self.last_frame``, what reward hooks difference against) is the simulator's
    second state buffer (``rsx_step_dev_flip``), not a clone."""
    KIND = None
    NORM_BOUNDS = 1.2
    _LEVER_ARM = 0.04

    def __init__(self, field_type, n_robots_blue, n_robots_yellow, time_step, num_envs, device=0,
                 keep_last_frame=True, max_episode_steps=None, auto_reset=True):
        import torch
        self._torch = torch
        self.num_envs = int(num_envs)
        self.device = torch.device("cuda", int(device))
        self.time_step = time_step
        self.n_robots_blue, self.n_robots_yellow = n_robots_blue, n_robots_yellow
        self.field_type = field_type
        self.max_episode_steps = max_episode_steps
        self.auto_reset = auto_reset
        self.sim = _lib.Sim(self.KIND, field_type, n_robots_blue, n_robots_yellow, int(time_step * 1000),
                            self.num_envs, int(device))
        self.field = Field(**self.sim.get_field_params())
        self.max_pos = max(self.field.width / 2, self.field.length / 2 + self.field.penalty_length)
        self.max_v = (self.field.rbt_motor_max_rpm / 60) * 2 * np.pi * self.field.rbt_wheel_radius
        self.max_w = float(np.rad2deg(self.max_v / self._LEVER_ARM))
        n = n_robots_blue + n_robots_yellow
        self.keep_last_frame = keep_last_frame
        if keep_last_frame:   # two state buffers that trade places every step
            cur, oth = self.sim.state_buffers()
            self.frame = VecFrame(cur, n_robots_blue, n_robots_yellow, self.KIND)
            self._other = VecFrame(oth, n_robots_blue, n_robots_yellow, self.KIND)
        else:
            self.frame = VecFrame(self.sim.state_tensor(), n_robots_blue, n_robots_yellow, self.KIND)
            self._other = None
        self.commands = self.sim.cmds_tensor().view(n, self.sim.cmd_dim, self.num_envs)
        self.last_frame = None
        self.steps = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        self._device_placement = None   # whether _get_initial_positions() returns device tensors (learnt in _place())
        self._graph_mode = False

    def enable_graph_capture(self):
        """Make ``step()`` capturable into a ``torch.cuda.CUDAGraph`` (hooks + raw step + device-side auto-reset as one graph, the
        way ``VecFusedEnv.enable_graph_capture`` does for the fused tasks).  What changes: ``self.frame`` / ``self.last_frame`` stay
        the SAME two buffers for good — the previous frame is kept by one device copy per step instead of by trading the buffers'
        roles (a replayed graph would keep writing the buffer that was current when it was captured; ``rsx_step_dev_flip`` refuses
        to be captured for that reason).  Requires placements that stay on the device: call after a ``reset()`` whose
        ``_get_initial_positions`` returned device tensors (or with ``auto_reset=False``).  The raw step holds no host state and
        draws nothing, so replays need no counter: results are those of the eager calls."""
        if self.auto_reset and self._device_placement is not True:
            raise RuntimeError("enable_graph_capture() needs device-side placement: reset() first, with _get_initial_positions() returning "
                               "device tensors (host-array placement costs a synchronisation per episode end and cannot be captured)")
        self._graph_mode = True
        _lib.drop_pending_hip_error()   # an earlier capture attempt that was refused and aborted leaves one behind (rsx.h)
        if self.keep_last_frame and self.last_frame is None:
            self.last_frame = self._other
            self._other.state.copy_(self.frame.state)
        return self

    def _stream(self):
        return self._torch.cuda.current_stream(self.device).cuda_stream

    def _place(self, env_mask):
        """teleport the envs selected by env_mask ([B] bool device tensor, host array or None = all)"""
        torch = self._torch
        ball, blue, yellow = self._get_initial_positions()
        self._device_placement = isinstance(ball, torch.Tensor)   # learnt from the value at hand: no extra call of the hook
        if self._device_placement:             # device placement: stream-ordered, no host copy, no sync
            m = None if env_mask is None else torch.as_tensor(env_mask, device=self.device)
            self.sim.reset_dev(ball, blue, yellow, m, self._stream())
        else:                                  # host arrays (robosim.reset format): PCIe + synchronisation
            m = env_mask.cpu().numpy() if isinstance(env_mask, torch.Tensor) else env_mask
            self.sim.reset(ball, blue, yellow, m, self._stream())

    def step(self, action):
        torch = self._torch
        self.steps += 1
        self.commands.zero_()
        self._get_commands(action)          # fills self.commands
        if self.keep_last_frame and self._graph_mode:
            self._other.state.copy_(self.frame.state)     # fixed roles (see enable_graph_capture): the previous frame by copy
            self.last_frame = self._other
            self.sim.step_dev(self._stream())
        elif self.keep_last_frame:
            self.sim.step_dev_flip(self._stream())
            self.frame, self._other = self._other, self.frame
            self.last_frame = self._other
        else:
            self.sim.step_dev(self._stream())
        obs = self._frame_to_observations()
        reward, done = self._calculate_reward_and_done()
        done = done.to(torch.bool)
        truncated = self.steps >= self.max_episode_steps if self.max_episode_steps else torch.zeros_like(done)   # a fresh tensor per call
        info = {}
        if self.auto_reset:
            ended = done | truncated
            info["final_obs"] = obs.clone()     # the hook may hand out a buffer it reuses
            if self._device_placement is None:
                # step() before any reset(): not yet known what the placement hook returns.  The first placement of envs that
                # really ended tells — the hook is never called just to look at its return type (a user's hook draws from its
                # own random stream: runs that do and do not reset() first must see the same stream)
                if bool(ended.any()):
                    self._place(ended)
                    self.steps.masked_fill_(ended, 0)
                    obs = torch.where(ended[:, None], self._frame_to_observations(), obs)
            elif self._device_placement:
                # device placement: stream-ordered and masked on the device, no host round trip — every step
                self._place(ended)
                self.steps.masked_fill_(ended, 0)
                # the placement wrote the CURRENT buffer; reset envs get their first observation
                obs = torch.where(ended[:, None], self._frame_to_observations(), obs)
            elif bool(ended.any()):
                # host-array placement (robosim.reset format) costs a synchronisation and a state round trip: only
                # when an episode really ended (`ended.any()` is itself one small read-back per step on this path)
                self._place(ended)
                self.steps.masked_fill_(ended, 0)
                obs = torch.where(ended[:, None], self._frame_to_observations(), obs)
        return obs, reward, done, truncated, info

    def reset(self, env_mask=None):
        """(Re)place the envs selected by ``env_mask`` (bool device tensor or host array, default all)."""
        self._place(env_mask)
        if env_mask is None:
            self.steps.zero_()
        else:
            self.steps.masked_fill_(self._torch.as_tensor(env_mask, device=self.device).to(self._torch.bool), 0)
        # no frame precedes a reset — except in graph mode, where the two buffers keep their roles for good (enable_graph_capture): a
        # replayed step() refreshes the previous frame on the device, and the attribute a caller reads must not depend on whether a
        # reset() happened to come between the capture and the replay
        self.last_frame = self._other if (self._graph_mode and self.keep_last_frame) else None
        return self._frame_to_observations(), {}

    def close(self):
        self.sim.close()

    # ---- hooks ----
    def _get_commands(self, action):
        """write the commands of this step into self.commands ([n_robots, C, B])"""
        raise NotImplementedError

    def _frame_to_observations(self):
        """returns the [B, obs_dim] observation tensor from self.frame"""
        raise NotImplementedError

    def _calculate_reward_and_done(self):
        """returns ([B] reward, [B] done) tensors from self.frame / self.last_frame"""
        raise NotImplementedError

    def _get_initial_positions(self):
        """returns ball [B,4], blue [B,nb,3], yellow [B,ny,3] — float32 device tensors (no host
        traffic) or host arrays (the robosim.reset format)"""
        raise NotImplementedError

    # ---- normalisation helpers (vss_gym_base.py:213-220) ----
    def norm_pos(self, pos):
        return self._torch.clamp(pos / self.max_pos, -self.NORM_BOUNDS, self.NORM_BOUNDS)

    def norm_v(self, v):
        return self._torch.clamp(v / self.max_v, -self.NORM_BOUNDS, self.NORM_BOUNDS)

    def norm_w(self, w):
        return self._torch.clamp(w / self.max_w, -self.NORM_BOUNDS, self.NORM_BOUNDS)


class VecVSSBaseEnv(_VecBaseEnv):
    KIND = _lib.KIND_VSS
    _LEVER_ARM = 0.04


class VecSSLBaseEnv(_VecBaseEnv):
    KIND = _lib.KIND_SSL
    _LEVER_ARM = 0.095
