"""Adapters between the gym-style envs and the native simulator (librsx_hip via rsoccer_amd.robosim)."""
