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
