"""World snapshot parsed from a simulator state vector.

Layouts (rsoccer_gym/Entities/Frame.py:20-47 and :55-92): ``[ball x, y, z, v_x, v_y]`` followed
by one block per robot, blue ids first then yellow ids — 6 values for VSS
(x, y, theta, v_x, v_y, v_theta) and 11 for SSL (+ infrared, v_wheel0..3).
Units: seconds, m, m/s, degrees, degrees/s; origin at the field centre.

``parse()`` keeps the state vector (``frame.state``) and builds the ``Ball`` / ``Robot`` records only when
``ball`` / ``robots_blue`` / ``robots_yellow`` are first read: the task classes of this package compute
observations and rewards from the vector in a few array operations (the reference's per-value scalar calls
are what caps it at ~4.9 k steps/s), while code written against the records — user tasks, the renderer —
sees exactly the reference's object graph.
"""
from typing import Dict

from rsoccer_amd.Entities.Ball import Ball
from rsoccer_amd.Entities.Robot import Robot

_VSS_BLOCK = ("x", "y", "theta", "v_x", "v_y", "v_theta")
_SSL_BLOCK = _VSS_BLOCK + ("infrared", "v_wheel0", "v_wheel1", "v_wheel2", "v_wheel3")
_RECORDS = ("ball", "robots_blue", "robots_yellow")


class Frame:
    """One world snapshot; s, m, m/s, degrees, degrees/s; origin = field centre."""

    _block = ()
    state = None          # the vector parse() was given (None for a frame assembled by hand)
    _counts = (0, 0)

    ball: Ball
    robots_blue: Dict[int, Robot]
    robots_yellow: Dict[int, Robot]

    def __getattr__(self, name):
        # only reached when the attribute is not there yet: the three records are built on first use
        if name in _RECORDS:
            self._materialise()
            return self.__dict__[name]
        raise AttributeError(name)

    def _materialise(self):
        d = self.__dict__
        if self.state is None:    # a frame assembled by hand (placements): empty records, what is already set stays
            d.setdefault("ball", Ball())
            d.setdefault("robots_blue", {})
            d.setdefault("robots_yellow", {})
            return
        d["ball"], d["robots_blue"], d["robots_yellow"] = Ball(), {}, {}
        block = self._block
        values = self.state.tolist() if hasattr(self.state, "tolist") else list(self.state)
        state = self.state
        ball = d["ball"]
        # records hold the entries of the array itself (numpy scalars), as the reference's parse() does
        ball.x, ball.y, ball.z, ball.v_x, ball.v_y = (state[i] for i in range(5))
        width = len(block)
        n_blues, n_yellows = self._counts
        for team, count, first in ((d["robots_blue"], n_blues, 0), (d["robots_yellow"], n_yellows, n_blues)):
            for i in range(count):
                base = 5 + width * (first + i)
                robot = Robot(id=i)
                for k, name in enumerate(block):
                    robot.__dict__[name] = bool(values[base + k]) if name == "infrared" else state[base + k]
                team[i] = robot

    def parse(self, state, n_blues=3, n_yellows=3):
        if not self._block:
            raise NotImplementedError("use FrameVSS or FrameSSL")
        for name in _RECORDS:
            self.__dict__.pop(name, None)
        self.state = state
        self._counts = (n_blues, n_yellows)
        return self


class FrameVSS(Frame):
    _block = _VSS_BLOCK


class FrameSSL(Frame):
    _block = _SSL_BLOCK
