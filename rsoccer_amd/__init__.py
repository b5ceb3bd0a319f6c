"""rsoccer_amd — MI355X-native vectorised rSoccer step engine."""
__version__ = "0.1.0"
