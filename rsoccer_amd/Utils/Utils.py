"""Ornstein-Uhlenbeck exploration noise for the robots the agent does not control
(rsoccer_gym/Utils/Utils.py:5-29): x <- x + theta (mu - x) dt + sigma sqrt(dt) N(0, 1), with
mu / sigma derived from the action box; draws come from numpy's global generator like the
reference, so ``np.random.seed`` reproduces its streams."""
import numpy as np


class OrnsteinUhlenbeckAction:
    def __init__(self, action_space, theta=0.17, dt=0.025, x0=None):
        self.theta, self.dt, self.x0 = theta, dt, x0
        self.mu = (action_space.high + action_space.low) / 2
        self.sigma = (action_space.high - self.mu) / 2
        self.reset()

    def reset(self):
        self.x_prev = np.zeros_like(self.mu) if self.x0 is None else self.x0

    def sample(self):
        drift = self.theta * (self.mu - self.x_prev) * self.dt
        shock = self.sigma * np.sqrt(self.dt) * np.random.normal(size=self.mu.shape)
        self.x_prev = self.x_prev + drift + shock
        return self.x_prev

    def __repr__(self):
        return f"OrnsteinUhlenbeckActionNoise(mu={self.mu}, sigma={self.sigma})"


class OrnsteinUhlenbeckBank:
    """Several :class:`OrnsteinUhlenbeckAction` processes of the same shape advanced by ONE draw and three array
    operations per step.  Same numbers as sampling them one after the other: numpy's global generator hands out its
    normals sequentially, so ``normal(size=(k, n))`` is the concatenation of k draws of size n, and every element sees
    the same float64 operations in the same order (``x + theta (mu - x) dt + sigma sqrt(dt) N``).  The processes stay
    the owners of their state (``x_prev``): it is read before and written after every bank step, so sampling one of
    them by hand in between, or resetting it, is honoured."""

    def __init__(self, processes):
        self.processes = list(processes)
        p = self.processes[0]
        if any(q.theta != p.theta or q.dt != p.dt or q.mu.shape != p.mu.shape for q in self.processes):
            raise ValueError("the bank needs processes of one shape, theta and dt")
        self.theta, self.dt = p.theta, p.dt
        self.mu = np.stack([np.asarray(q.mu, dtype=np.float64) for q in self.processes])
        self.shock = np.stack([q.sigma * np.sqrt(q.dt) for q in self.processes])   # float64, as in sample()
        self._x = np.zeros(self.mu.shape, dtype=np.float64)
        self._rows = [None] * len(self.processes)

    def sample(self):
        """advance every process; returns the [k, n] array of their new values (row i is also processes[i].x_prev)"""
        x, rows = self._x, self._rows
        for i, q in enumerate(self.processes):
            if q.x_prev is not rows[i]:          # reset, or sampled by hand, since the last bank step
                x[i] = q.x_prev
        new = x + self.theta * (self.mu - x) * self.dt + self.shock * np.random.normal(size=self.mu.shape)
        self._x = new
        for i, q in enumerate(self.processes):
            rows[i] = q.x_prev = new[i]
        return new
