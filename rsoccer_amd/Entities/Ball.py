"""Ball record — same attribute names as the reference (rsoccer_gym/Entities/Ball.py:3-10)."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class Ball:
    x: Optional[float] = None
    y: Optional[float] = None
    z: Optional[float] = None
    v_x: float = 0.0
    v_y: float = 0.0
    v_z: float = 0.0
