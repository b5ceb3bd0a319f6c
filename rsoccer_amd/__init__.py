"""rsoccer_amd — MI355X-native vectorised rSoccer step engine.

Importing the package registers the environment ids of the reference registry
(rsoccer_gym/__init__.py:3-30) that this build implements:

    VSS-v0                     VSSEnv                       1200 steps
    SSLStaticDefenders-v0      SSLHWStaticDefendersEnv      1000 steps
    SSLDribbling-v0            SSLHWDribblingEnv            4800 steps
    SSLContestedPossession-v0  SSLContestedPossessionEnv    1200 steps
    SSLPassEndurance-v0        SSLPassEnduranceEnv          1200 steps

``rsoccer_amd.make(id)`` returns the single-environment, reference-shaped object (hooks in
Python, physics on the GPU).  ``rsoccer_amd.make_vec(id, num_envs)`` returns the batched, fully fused version (the classes of
``rsoccer_amd.vec``).
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

# registered id -> the fused batched class of rsoccer_amd.vec that steps it (one kernel launch per step for the whole batch)
VECTOR_CLASSES = {
    "VSS-v0": "VecVSSEnv",
    "SSLStaticDefenders-v0": "VecSSLStaticDefendersEnv",
    "SSLDribbling-v0": "VecSSLDribblingEnv",
    "SSLContestedPossession-v0": "VecSSLContestedPossessionEnv",
    "SSLPassEndurance-v0": "VecSSLPassEnduranceEnv",
}


def make_vec(id, num_envs, **kwargs):
    """``num_envs`` copies of a registered task on one GPU — the batched counterpart of ``make(id)`` (the shape of
    ``gymnasium.make_vec(id, num_envs=...)``): the fused class of :mod:`rsoccer_amd.vec` for that id, with the registry's episode
    limit unless ``max_episode_steps`` says otherwise.  Further keywords go to the class: ``device``, ``seed``, ``env_id_base``
    (global id of env 0: shards of one population over several GPUs), ``max_episode_steps``."""
    if id not in VECTOR_CLASSES:
        raise KeyError(f"no batched environment for id {id!r}; registered: {sorted(VECTOR_CLASSES)}")
    from rsoccer_amd import vec
    return getattr(vec, VECTOR_CLASSES[id])(int(num_envs), **kwargs)


__all__ = ["make", "make_vec", "register", "registry", "VECTOR_CLASSES", "__version__"]
