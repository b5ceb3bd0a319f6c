"""``robosim``-compatible classes on top of the HIP step engine.

The reference imports the third-party module ``robosim`` (rc-robosim) at
rsoccer_gym/Simulators/rsim.py:2 and uses exactly this surface: ``VSS(...)`` / ``SSL(...)``
constructors (rsim.py:116-124, :169-177), ``step(cmds)`` (:102, :155), ``get_state()``
(:105, :158), ``reset(ball, blue, yellow)`` (:38), ``get_field_params()`` (:50) and destruction
by ``del`` (:41).  ``import rsoccer_amd.robosim as robosim`` is the drop-in: same arguments, same
array layouts (float64), one environment per object — executed by librsx_hip.so on the GPU
(batch of one).  For throughput use :mod:`rsoccer_amd.vec` instead.
"""
import numpy as np

from . import _lib


class _Sim:
    _KIND = None

    def __init__(self, field_type, n_robots_blue, n_robots_yellow, time_step_ms,
                 ball_pos, blue_robots_pos, yellow_robots_pos, device_id=0):
        self._sim = _lib.Sim(self._KIND, int(field_type), int(n_robots_blue), int(n_robots_yellow),
                             int(time_step_ms), 1, device_id)
        self.n_robots_blue = int(n_robots_blue)
        self.n_robots_yellow = int(n_robots_yellow)
        self._n_cmd = self._sim.n_robots * self._sim.cmd_dim
        self._state = None   # the state the last step() brought home, until get_state() hands it out
        self.reset(np.asarray(ball_pos, dtype=np.float64),
                   np.asarray(blue_robots_pos, dtype=np.float64),
                   np.asarray(yellow_robots_pos, dtype=np.float64))

    def step(self, commands):
        """commands: float64 [n_robots, 2] (VSS) or [n_robots, 8] (SSL).  The new state comes home in the same call
        (rsx_step_state: the reference always follows step() with get_state(), rsim.py:102,105)."""
        c = commands
        if not (type(c) is np.ndarray and c.dtype == np.float64 and c.flags.c_contiguous and c.size == self._n_cmd):
            c = np.ascontiguousarray(commands, dtype=np.float64)
            if c.size != self._n_cmd:
                raise ValueError(f"expected {self._n_cmd} command values, got {c.shape}")
        self._state = self._sim.step_state(c)[0]

    def get_state(self):
        s = self._state
        if s is None:
            s = self._sim.get_state()[0]
        else:
            self._state = None    # handed out once: every call returns a fresh array, like robosim
        return s

    def reset(self, ball_pos, blue_robots_pos, yellow_robots_pos):
        nb, ny = self.n_robots_blue, self.n_robots_yellow
        blue = np.asarray(blue_robots_pos, dtype=np.float64).reshape(nb, 3) if nb else None
        yellow = np.asarray(yellow_robots_pos, dtype=np.float64).reshape(ny, 3) if ny else None
        self._state = None
        self._sim.reset(np.asarray(ball_pos, dtype=np.float64).reshape(1, 4), blue, yellow)

    def get_field_params(self):
        return self._sim.get_field_params()

    def close(self):
        self._sim.close()

    def __del__(self):
        try:
            self._sim.close()
        except Exception:
            pass


class VSS(_Sim):
    _KIND = _lib.KIND_VSS


class SSL(_Sim):
    _KIND = _lib.KIND_SSL
