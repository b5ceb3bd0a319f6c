from rsoccer_amd.ssl.ssl_hw_challenge.contested_possession import SSLContestedPossessionEnv
from rsoccer_amd.ssl.ssl_hw_challenge.dribbling import SSLHWDribblingEnv
from rsoccer_amd.ssl.ssl_hw_challenge.pass_endurance import SSLPassEnduranceEnv
from rsoccer_amd.ssl.ssl_hw_challenge.static_defenders import SSLHWStaticDefendersEnv

__all__ = ["SSLHWStaticDefendersEnv", "SSLHWDribblingEnv", "SSLContestedPossessionEnv", "SSLPassEnduranceEnv"]
