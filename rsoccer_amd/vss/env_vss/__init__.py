from rsoccer_amd.vss.env_vss.vss_gym import VSSEnv

__all__ = ["VSSEnv"]
