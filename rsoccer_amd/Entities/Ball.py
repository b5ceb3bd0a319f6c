"""``Ball`` — rsoccer_gym/Entities/Ball.py:3-10: position (m, field-centre origin) and velocity (m/s); every field optional / zero by
default, so that tasks build partial records for placements (vss_gym.py:200-206)."""
from rsoccer_amd.Entities.records import OptFloat, record

Ball = record("Ball", __name__, [
    ("x", OptFloat, None), ("y", OptFloat, None), ("z", OptFloat, None),   # m
    ("v_x", float, 0.0), ("v_y", float, 0.0), ("v_z", float, 0.0),         # m/s
], "Ball pose and velocity (m, m/s).")
