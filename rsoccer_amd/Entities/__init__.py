"""Records (Ball, Robot, Field) and state-vector parsers (Frame, FrameVSS, FrameSSL)."""
from rsoccer_amd.Entities.records import Ball, Field, Robot
from rsoccer_amd.Entities.Frame import Frame, FrameSSL, FrameVSS

__all__ = ["Ball", "Field", "Frame", "FrameSSL", "FrameVSS", "Robot"]
