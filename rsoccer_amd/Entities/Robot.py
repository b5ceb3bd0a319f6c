"""``Robot`` — rsoccer_gym/Entities/Robot.py:4-23: the state a simulator reports (Frame.py:30-47 / :64-92) and the command fields a
task fills in (rsim.py:92-101 reads v_wheel0/1 of a VSS robot, :129-153 the eight SSL columns) in one record."""
from typing import Optional

from rsoccer_amd.Entities.records import OptFloat, record

Robot = record("Robot", __name__, [
    ("yellow", Optional[bool], None), ("id", Optional[int], None),
    ("x", OptFloat, None), ("y", OptFloat, None), ("z", OptFloat, None),   # m
    ("theta", OptFloat, None),                                             # degrees
    ("v_x", float, 0), ("v_y", float, 0), ("v_theta", float, 0),           # state: m/s, deg/s; SSL command: robot-local m/s, rad/s
    ("kick_v_x", float, 0), ("kick_v_z", float, 0),                        # m/s
    ("dribbler", bool, False), ("infrared", bool, False), ("wheel_speed", bool, False),
    ("v_wheel0", float, 0), ("v_wheel1", float, 0), ("v_wheel2", float, 0), ("v_wheel3", float, 0),   # rad/s
], "Robot state as read from the simulator and the command fields a task fills in.")
