"""Nearest-neighbour index used by the random placement of the tasks
(rsoccer_gym/Utils/kdtree.py:21-88; call sites vss_gym.py:213-231, static_defenders.py:241-252).

``insert(point)`` / ``get_nearest(point) -> (nearest_point, distance)``.  The placement only
ever holds a couple of dozen points, so this is a flat list scanned linearly: the answers are
those of the reference's tree (exact nearest neighbour), without its recursion."""
import math


class KDTree:
    def __init__(self):
        self._points = []

    def insert(self, values):
        self._points.append(tuple(values))

    def get_nearest(self, values):
        if not self._points:
            raise ValueError("empty tree")
        best, best_d2 = None, math.inf
        for p in self._points:
            d2 = sum((a - b) ** 2 for a, b in zip(values, p))
            if d2 < best_d2:
                best, best_d2 = p, d2
        return best, math.sqrt(best_d2)

    def __len__(self):
        return len(self._points)
