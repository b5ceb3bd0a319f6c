from .Utils import OrnsteinUhlenbeckAction
from .kdtree import KDTree

__all__ = ["OrnsteinUhlenbeckAction", "KDTree"]
