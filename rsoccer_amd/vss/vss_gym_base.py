"""Base class of the single-environment VSS tasks (IEEE Very Small Size Soccer).

Same contract as the reference's ``VSSBaseEnv`` (rsoccer_gym/vss/vss_gym_base.py:19-220):
construct the simulator adapter, derive the normalisation constants, run the
``step()`` / ``reset()`` template and let a subclass supply four hooks —
``_get_commands``, ``_frame_to_observations``, ``_calculate_reward_and_done`` and
``_get_initial_positions_frame`` (:197-211).  Existing task classes written against the
reference work unchanged on top of this class; the physics behind ``self.rsim`` is the HIP step
engine.  This is the compatibility path (one env, one small GPU launch per step); training
throughput comes from :mod:`rsoccer_amd.vec`.
"""
from typing import List

import numpy as np

from rsoccer_amd import gymshim as gym
from rsoccer_amd.Entities import Frame, Robot
from rsoccer_amd.Simulators.rsim import RSimVSS


class VSSBaseEnv(gym.Env):
    # "human" (the reference's pygame window, Render/) is outside the step engine's scope
    metadata = {"render.modes": ["rgb_array"], "render_modes": ["rgb_array"],
                "render_fps": 60, "render.fps": 60}
    NORM_BOUNDS = 1.2
    _SIM_ADAPTER = RSimVSS
    _RENDER_VIEW = "VSS_VIEW"   # Render/field.py:189-201
    _LEVER_ARM = 0.04  # robot radius 0.0375 + wheel thickness 0.0025 (vss_gym_base.py:57-58)

    def __init__(self, field_type: int, n_robots_blue: int, n_robots_yellow: int, time_step: float,
                 render_mode=None, sim_backend=None):
        super().__init__()
        if render_mode not in (None, "rgb_array"):
            raise ValueError(f"render_mode={render_mode!r} is not supported (None or 'rgb_array'); the pygame "
                             "window of the reference is outside the scope of the step engine")
        self.render_mode = render_mode
        self.time_step = time_step
        self.rsim = self._SIM_ADAPTER(field_type=field_type, n_robots_blue=n_robots_blue,
                                      n_robots_yellow=n_robots_yellow,
                                      time_step_ms=int(self.time_step * 1000), backend=sim_backend)
        self.n_robots_blue = n_robots_blue
        self.n_robots_yellow = n_robots_yellow
        self.field_type = field_type
        self.field = self.rsim.get_field_params()
        # normalisers (vss_gym_base.py:52-58 / ssl_gym_base.py:53-59)
        self.max_pos = max(self.field.width / 2, (self.field.length / 2) + self.field.penalty_length)
        max_wheel_rad_s = (self.field.rbt_motor_max_rpm / 60) * 2 * np.pi
        self.max_v = max_wheel_rad_s * self.field.rbt_wheel_radius
        self.max_w = np.rad2deg(self.max_v / self._LEVER_ARM)
        self.frame: Frame = None
        self.last_frame: Frame = None
        self.steps = 0
        self.sent_commands = None

    # ---- template methods ----
    def step(self, action):
        self.steps += 1
        commands: List[Robot] = self._get_commands(action)
        self.rsim.send_commands(commands)
        self.sent_commands = commands
        self.last_frame, self.frame = self.frame, self.rsim.get_frame()
        observation = self._frame_to_observations()
        reward, done = self._calculate_reward_and_done()
        return observation, reward, done, False, {}

    def reset(self, *, seed=None, options=None):
        super().reset(seed=seed, options=options)
        self.steps = 0
        self.last_frame = None
        self.sent_commands = None
        self.rsim.reset(self._get_initial_positions_frame())
        self.frame = self.rsim.get_frame()
        obs = self._frame_to_observations()
        return obs, {}

    def render(self):
        """``rgb_array``: the current frame as uint8 [H, W, 3] in the reference's window geometry
        (numpy rasteriser, rsoccer_amd/Render/raster.py)."""
        if self.render_mode != "rgb_array":
            raise NotImplementedError("render() needs render_mode='rgb_array'")
        if getattr(self, "_raster", None) is None:
            from rsoccer_amd import Render
            self._raster = Render.FieldRaster(getattr(Render, self._RENDER_VIEW))
            self.window_size = self._raster.window_size
        return self._raster.draw(self.frame)

    def close(self):
        if self.rsim is not None:
            self.rsim.stop()
            self.rsim = None

    # ---- hooks a task implements ----
    def _get_commands(self, action):
        """action (as drawn from action_space) -> the List[Robot] commands of this step"""
        raise NotImplementedError

    def _frame_to_observations(self):
        """self.frame -> observation vector of observation_space"""
        raise NotImplementedError

    def _calculate_reward_and_done(self):
        """self.frame (and self.last_frame) -> (reward, done)"""
        raise NotImplementedError

    def _get_initial_positions_frame(self) -> Frame:
        """the Frame holding the start poses of an episode (reset)"""
        raise NotImplementedError

    # ---- observations in array form ----
    def _observation_plan(self, entries):
        """Compiles an observation layout into index / scale arrays for :meth:`_observe`.  ``entries``: one
        ``(state index, kind)`` per observation slot, kind in ``pos | v | w`` (normalised like norm_pos / norm_v /
        norm_w), ``sin | cos`` (of the heading in degrees at that index), ``flag`` (1 if non-zero else 0), ``sflag`` (1 if non-zero
        else -1), ``const`` (0: a slot the task fills itself)."""
        idx = np.array([i for i, _ in entries], dtype=np.intp)
        kinds = [k for _, k in entries]
        scale = {"pos": self.max_pos, "v": self.max_v, "w": self.max_w}
        den = np.array([scale.get(k, 1.0) for k in kinds], dtype=np.float64)
        pick = lambda kind: np.array([n for n, k in enumerate(kinds) if k == kind], dtype=np.intp)
        return {"idx": idx, "den": den, "sin": pick("sin"), "cos": pick("cos"), "flag": pick("flag"),
                "sflag": pick("sflag"), "const": pick("const"), "key": (self.max_pos, self.max_v, self.max_w)}

    def _observe(self, plan, state, lead=None):
        """the float32 observation vector of ``state`` — per slot exactly what the reference's scalar hooks compute
        (``np.clip(value / max, -1.2, 1.2)``, ``np.sin(np.deg2rad(theta))``), in a handful of array operations"""
        raw = state[plan["idx"]]
        out = raw / plan["den"]
        np.maximum(out, -self.NORM_BOUNDS, out=out)     # np.clip's two ufuncs, without its Python dispatch
        np.minimum(out, self.NORM_BOUNDS, out=out)
        if len(plan["sin"]):
            out[plan["sin"]] = np.sin(np.deg2rad(raw[plan["sin"]]))
            out[plan["cos"]] = np.cos(np.deg2rad(raw[plan["cos"]]))
        if len(plan["flag"]):
            out[plan["flag"]] = raw[plan["flag"]] != 0
        if len(plan["sflag"]):
            out[plan["sflag"]] = np.where(raw[plan["sflag"]] != 0, 1.0, -1.0)
        if len(plan["const"]):
            out[plan["const"]] = 0.0 if lead is None else lead
        return out.astype(np.float32)

    # ---- normalisation helpers ----
    def norm_pos(self, pos):
        return np.clip(pos / self.max_pos, -self.NORM_BOUNDS, self.NORM_BOUNDS)

    def norm_v(self, v):
        return np.clip(v / self.max_v, -self.NORM_BOUNDS, self.NORM_BOUNDS)

    def norm_w(self, w):
        return np.clip(w / self.max_w, -self.NORM_BOUNDS, self.NORM_BOUNDS)
