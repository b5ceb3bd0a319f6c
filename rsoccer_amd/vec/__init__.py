"""Batched environments: thousands of independent copies of a task on one GPU.

* :class:`VecVSSEnv`, :class:`VecSSLStaticDefendersEnv`, :class:`VecSSLDribblingEnv`,
  :class:`VecSSLContestedPossessionEnv`, :class:`VecSSLPassEnduranceEnv` — the five registered
  reference tasks with everything the reference does per step in Python (action ->
  commands, OU noise, physics, observation, reward, done, TimeLimit, reset placement) fused
  into one kernel launch per ``step()``.  Observations / rewards / flags are device tensors.
* :class:`VecVSSBaseEnv`, :class:`VecSSLBaseEnv` — the batched form of the subclass contract
  (``_get_commands`` / ``_frame_to_observations`` / ``_calculate_reward_and_done`` /
  ``_get_initial_positions``): hooks receive a :class:`VecFrame` whose fields are ``[B]``
  tensors viewing the simulator's SoA state, and run as torch ops on the device; TimeLimit and
  same-step auto-reset happen on the device (no host copy per step).
* :class:`VecScalarHookEnv` — unmodified scalar-hook task classes (the reference's contract,
  vss_gym_base.py:197-211), ``num_envs`` Python objects over ONE batched simulator.
"""
from rsoccer_amd.vec.fused import (VecFusedEnv, VecSSLContestedPossessionEnv, VecSSLDribblingEnv,
                                   VecSSLPassEnduranceEnv, VecSSLScrimmageEnv, VecSSLStaticDefendersEnv, VecVSSEnv)
from rsoccer_amd.vec.hooks import VecFrame, VecSSLBaseEnv, VecVSSBaseEnv
from rsoccer_amd.vec.scalar import VecScalarHookEnv

__all__ = ["VecFusedEnv", "VecVSSEnv", "VecSSLStaticDefendersEnv", "VecSSLDribblingEnv",
           "VecSSLContestedPossessionEnv", "VecSSLPassEnduranceEnv", "VecSSLScrimmageEnv", "VecFrame", "VecVSSBaseEnv", "VecSSLBaseEnv",
           "VecScalarHookEnv"]
