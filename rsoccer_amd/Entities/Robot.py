"""Robot record: pose / velocity as read from the simulator and the command fields the task
sets — attribute names of rsoccer_gym/Entities/Robot.py:4-23.  Units: m, m/s, degrees,
degrees/s; wheel speeds rad/s; v_x / v_y / v_theta of an SSL *command* are robot-local m/s and
rad/s."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class Robot:
    yellow: Optional[bool] = None
    id: Optional[int] = None
    x: Optional[float] = None
    y: Optional[float] = None
    z: Optional[float] = None
    theta: Optional[float] = None
    v_x: float = 0
    v_y: float = 0
    v_theta: float = 0
    kick_v_x: float = 0
    kick_v_z: float = 0
    dribbler: bool = False
    infrared: bool = False
    wheel_speed: bool = False
    v_wheel0: float = 0
    v_wheel1: float = 0
    v_wheel2: float = 0
    v_wheel3: float = 0
