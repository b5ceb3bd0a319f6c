"""World snapshot parsed from a simulator state vector.

Layouts (rsoccer_gym/Entities/Frame.py:20-47 and :55-92): ``[ball x, y, z, v_x, v_y]`` followed
by one block per robot, blue ids first then yellow ids — 6 values for VSS
(x, y, theta, v_x, v_y, v_theta) and 11 for SSL (+ infrared, v_wheel0..3).
Units: seconds, m, m/s, degrees, degrees/s; origin at the field centre.
"""
from typing import Dict

from rsoccer_amd.Entities.records import Ball, Robot

_VSS_BLOCK = ("x", "y", "theta", "v_x", "v_y", "v_theta")
_SSL_BLOCK = _VSS_BLOCK + ("infrared", "v_wheel0", "v_wheel1", "v_wheel2", "v_wheel3")


class Frame:
    """One world snapshot; s, m, m/s, degrees, degrees/s; origin = field centre."""

    _block = ()

    def __init__(self):
        self.ball: Ball = Ball()
        self.robots_blue: Dict[int, Robot] = {}
        self.robots_yellow: Dict[int, Robot] = {}

    def parse(self, state, n_blues=3, n_yellows=3):
        block = self._block
        if not block:
            raise NotImplementedError("use FrameVSS or FrameSSL")
        ball = self.ball
        ball.x, ball.y, ball.z, ball.v_x, ball.v_y = (state[i] for i in range(5))
        width = len(block)
        for team, count, first in ((self.robots_blue, n_blues, 0), (self.robots_yellow, n_yellows, n_blues)):
            for i in range(count):
                base = 5 + width * (first + i)
                robot = Robot(id=i)
                for k, name in enumerate(block):
                    value = state[base + k]
                    setattr(robot, name, bool(value) if name == "infrared" else value)
                team[i] = robot
        return self


class FrameVSS(Frame):
    _block = _VSS_BLOCK


class FrameSSL(Frame):
    _block = _SSL_BLOCK
