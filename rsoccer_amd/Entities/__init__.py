"""Records (Ball, Robot, Field) and state-vector parsers (Frame, FrameVSS, FrameSSL)."""
from rsoccer_amd.Entities.Ball import Ball
from rsoccer_amd.Entities.Field import Field
from rsoccer_amd.Entities.Robot import Robot
from rsoccer_amd.Entities.Frame import Frame, FrameSSL, FrameVSS

__all__ = ["Ball", "Field", "Frame", "FrameSSL", "FrameVSS", "Robot"]
