"""rsoccer_amd — MI355X-native vectorised rSoccer step engine.

Importing the package registers the environment ids of the reference registry
(rsoccer_gym/__init__.py:3-30) that this build implements:

    VSS-v0                     VSSEnv                       1200 steps
    SSLStaticDefenders-v0      SSLHWStaticDefendersEnv      1000 steps
    SSLDribbling-v0            SSLHWDribblingEnv            4800 steps
    SSLContestedPossession-v0  SSLContestedPossessionEnv    1200 steps
    SSLPassEndurance-v0        SSLPassEnduranceEnv          1200 steps

``rsoccer_amd.make(id)`` returns the single-environment, reference-shaped object (hooks in
Python, physics on the GPU).  ``rsoccer_amd.vec`` holds the batched, fully fused versions.
"""
__version__ = "0.1.0"

from rsoccer_amd.gymshim import make, register, registry  # noqa: E402

register(id="VSS-v0", entry_point="rsoccer_amd.vss.env_vss:VSSEnv", max_episode_steps=1200)
register(id="SSLStaticDefenders-v0",
         entry_point="rsoccer_amd.ssl.ssl_hw_challenge.static_defenders:SSLHWStaticDefendersEnv",
         kwargs={"field_type": 2}, max_episode_steps=1000)
register(id="SSLDribbling-v0", entry_point="rsoccer_amd.ssl.ssl_hw_challenge.dribbling:SSLHWDribblingEnv",
         max_episode_steps=4800)
register(id="SSLContestedPossession-v0",
         entry_point="rsoccer_amd.ssl.ssl_hw_challenge.contested_possession:SSLContestedPossessionEnv",
         max_episode_steps=1200)
register(id="SSLPassEndurance-v0", entry_point="rsoccer_amd.ssl.ssl_hw_challenge:SSLPassEnduranceEnv",
         max_episode_steps=1200)

__all__ = ["make", "register", "registry", "__version__"]
