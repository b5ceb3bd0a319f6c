from rsoccer_amd.ssl.ssl_hw_challenge.static_defenders import SSLHWStaticDefendersEnv

__all__ = ["SSLHWStaticDefendersEnv"]
