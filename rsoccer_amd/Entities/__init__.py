from .Ball import Ball
from .Field import Field
from .Frame import Frame, FrameSSL, FrameVSS
from .Robot import Robot

__all__ = ["Ball", "Field", "Frame", "FrameSSL", "FrameVSS", "Robot"]
