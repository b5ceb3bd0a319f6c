"""Plain records exchanged between the tasks and the simulator adapters.

Attribute names, defaults and order are those of the reference's dataclasses
(rsoccer_gym/Entities/Ball.py:3-10, Robot.py:4-23, Field.py:3-21) — task code reads and writes
them by name and ``Field(**simulator.get_field_params())`` needs exactly the 17 keys.  The
classes are built from field tables (one place to see the layout, units alongside).
"""
from dataclasses import field, make_dataclass
from typing import Optional

from rsoccer_amd._lib import FIELD_KEYS

_opt = Optional[float]


def _record(name, spec, doc):
    cls = make_dataclass(name, [(n, t, field(default=d)) for n, t, d in spec])
    cls.__doc__ = doc
    cls.__module__ = __name__
    return cls


Ball = _record("Ball", [
    ("x", _opt, None), ("y", _opt, None), ("z", _opt, None),        # m, field-centre origin
    ("v_x", float, 0.0), ("v_y", float, 0.0), ("v_z", float, 0.0),  # m/s
], "Ball pose and velocity (m, m/s).")

Robot = _record("Robot", [
    ("yellow", Optional[bool], None), ("id", Optional[int], None),
    ("x", _opt, None), ("y", _opt, None), ("z", _opt, None),         # m
    ("theta", _opt, None),                                            # degrees
    ("v_x", float, 0), ("v_y", float, 0), ("v_theta", float, 0),     # state: m/s, deg/s; SSL command: robot-local m/s, rad/s
    ("kick_v_x", float, 0), ("kick_v_z", float, 0),                  # m/s
    ("dribbler", bool, False), ("infrared", bool, False), ("wheel_speed", bool, False),
    ("v_wheel0", float, 0), ("v_wheel1", float, 0), ("v_wheel2", float, 0), ("v_wheel3", float, 0),  # rad/s
], "Robot state as read from the simulator and the command fields a task fills in.")

Field = make_dataclass("Field", [(k, float) for k in FIELD_KEYS])
Field.__doc__ = "Field and robot geometry as returned by get_field_params(): " + ", ".join(FIELD_KEYS) + "."
Field.__module__ = __name__
