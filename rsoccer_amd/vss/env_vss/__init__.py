"""VSS-v0 task."""
from .vss_gym import VSSEnv

__all__ = ["VSSEnv"]
