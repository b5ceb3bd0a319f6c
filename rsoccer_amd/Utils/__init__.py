from .Utils import OrnsteinUhlenbeckAction, OrnsteinUhlenbeckBank
from .kdtree import KDTree

__all__ = ["OrnsteinUhlenbeckAction", "OrnsteinUhlenbeckBank", "KDTree"]
