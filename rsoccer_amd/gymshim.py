"""Gymnasium surface used by the environments.

When the real ``gymnasium`` package is importable it is used unchanged (``gym.Env``,
``gym.spaces.Box``, ``register`` / ``make`` with its TimeLimit wrapper).  It is not part of this
image, so a minimal stand-in with the same call signatures is provided: ``Env`` (seeded
``np_random``), ``spaces.Box``, ``TimeLimit`` and a registry — enough for
``make("VSS-v0")`` / ``reset()`` / ``step()`` to behave as README.md:116-133 shows.
"""
import importlib

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import gymnasium as _gym
    HAVE_GYMNASIUM = True
except Exception:  # ImportError and partial installs alike
    _gym = None
    HAVE_GYMNASIUM = False


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, dtype=self.dtype).reshape(self.shape)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, dtype=self.dtype).reshape(self.shape)
        self._rng = np.random.default_rng(seed)

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"


class _Spaces:
    Box = _Box


class _Env:
    metadata = {}
    render_mode = None
    action_space = None
    observation_space = None
    np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None or self.np_random is None:
            self.np_random = np.random.default_rng(seed)
        return None

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class _TimeLimit:
    """Truncates an episode after ``max_episode_steps`` steps (what gymnasium.make wraps the
    registered ids in — rsoccer_gym/__init__.py:4,11,17,23,29)."""

    def __init__(self, env, max_episode_steps):
        self.env = env
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = 0

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            truncated = True
        return obs, reward, terminated, truncated, info

    def close(self):
        return self.env.close()


_REGISTRY = {}


def _register(id, entry_point, max_episode_steps=None, kwargs=None, **_ignored):
    _REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, kwargs=dict(kwargs or {}))


def _make(id, **kwargs):
    try:
        spec = _REGISTRY[id]
    except KeyError:
        raise KeyError(f"unknown environment id {id!r}; registered: {sorted(_REGISTRY)}") from None
    module, _, attr = spec["entry_point"].partition(":")
    cls = getattr(importlib.import_module(module), attr)
    env = cls(**{**spec["kwargs"], **kwargs})
    if spec["max_episode_steps"]:
        env = _TimeLimit(env, spec["max_episode_steps"])
    return env


if HAVE_GYMNASIUM:  # pragma: no cover
    Env = _gym.Env
    spaces = _gym.spaces
    register = _gym.register
    make = _gym.make
    TimeLimit = _gym.wrappers.TimeLimit
    registry = _gym.registry   # the ids live in gymnasium's own registry
else:
    Env = _Env
    spaces = _Spaces
    register = _register
    make = _make
    TimeLimit = _TimeLimit
    registry = _REGISTRY
