"""A stand-in for the third-party ``gymnasium`` package, written to a temporary directory by tests/test_gymnasium_branch.py.

The image has no gymnasium; rsoccer_amd/gymshim.py takes another branch when it is importable (the real ``Env``, ``spaces.Box``,
``register`` / ``make`` / ``registry`` and the ``TimeLimit`` wrapper instead of its own minimal ones).  This module holds the sources
of a package with the public shape of gymnasium 0.29 / 1.x that the reference uses (rsoccer_gym/__init__.py:1-30, README.md:116-133:
``register(id=, entry_point=, kwargs=, max_episode_steps=)``, ``make(id, **kwargs)``, ``Env.reset(seed=, options=)`` seeding
``np_random``, ``registry`` = id -> ``EnvSpec``, ``make`` wrapping in ``wrappers.TimeLimit``) so that the branch runs under test.
TEST INFRASTRUCTURE; nothing in the product imports it."""

FILES = {
    "gymnasium/__init__.py": '''
from gymnasium.core import Env, Wrapper
from gymnasium import spaces, wrappers
from gymnasium.envs.registration import register, make, registry, spec, EnvSpec
__version__ = "0.29.1+stub"
''',
    "gymnasium/core.py": '''
import numpy as np


class Env:
    metadata = {"render_modes": []}
    render_mode = None
    spec = None
    action_space = None
    observation_space = None
    _np_random = None
    _np_random_seed = None

    def step(self, action):
        raise NotImplementedError

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.default_rng(seed)
            self._np_random_seed = seed

    def render(self):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.default_rng()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()
        return False


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def spec(self):
        return self.env.spec

    @property
    def action_space(self):
        return self.env.action_space

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def render_mode(self):
        return self.env.render_mode

    @property
    def np_random(self):
        return self.env.np_random

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def step(self, action):
        return self.env.step(action)

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def render(self):
        return self.env.render()

    def close(self):
        return self.env.close()
''',
    "gymnasium/spaces.py": '''
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.broadcast(np.asarray(low), np.asarray(high)).shape
        super().__init__(shape, dtype, seed)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
''',
    "gymnasium/wrappers.py": '''
from gymnasium.core import Wrapper


class OrderEnforcing(Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self._has_reset = False

    def step(self, action):
        if not self._has_reset:
            raise RuntimeError("Cannot call env.step() before calling env.reset()")
        return self.env.step(action)

    def reset(self, **kwargs):
        self._has_reset = True
        return self.env.reset(**kwargs)


class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps):
        super().__init__(env)
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        observation, reward, terminated, truncated, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            truncated = True
        return observation, reward, terminated, truncated, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)
''',
    "gymnasium/envs/__init__.py": "from gymnasium.envs.registration import register, make, registry, spec\n",
    "gymnasium/envs/registration.py": '''
import copy
import dataclasses
import importlib
import re

ENV_ID_RE = re.compile(r"^(?:(?P<namespace>[\\w:-]+)\\/)?(?:(?P<name>[\\w:.-]+?))(?:-v(?P<version>\\d+))?$")


@dataclasses.dataclass
class EnvSpec:
    id: str
    entry_point: object = None
    reward_threshold: object = None
    nondeterministic: bool = False
    max_episode_steps: object = None
    order_enforce: bool = True
    disable_env_checker: bool = False
    kwargs: dict = dataclasses.field(default_factory=dict)
    additional_wrappers: tuple = ()
    vector_entry_point: object = None


registry = {}


def register(id, entry_point=None, reward_threshold=None, nondeterministic=False, max_episode_steps=None, order_enforce=True,
             disable_env_checker=False, additional_wrappers=(), vector_entry_point=None, kwargs=None):
    assert entry_point is not None or vector_entry_point is not None
    if not ENV_ID_RE.fullmatch(id):
        raise ValueError(f"Malformed environment ID: {id}")
    registry[id] = EnvSpec(id=id, entry_point=entry_point, reward_threshold=reward_threshold, nondeterministic=nondeterministic,
                           max_episode_steps=max_episode_steps, order_enforce=order_enforce, disable_env_checker=disable_env_checker,
                           kwargs=dict(kwargs or {}), additional_wrappers=tuple(additional_wrappers), vector_entry_point=vector_entry_point)


def spec(env_id):
    if env_id not in registry:
        raise KeyError(f"No registered env with id: {env_id}")
    return registry[env_id]


def load_env_creator(name):
    mod_name, attr_name = name.split(":")
    return getattr(importlib.import_module(mod_name), attr_name)


def make(id, max_episode_steps=None, disable_env_checker=None, **kwargs):
    from gymnasium import wrappers
    env_spec = id if isinstance(id, EnvSpec) else spec(id)
    creator = env_spec.entry_point if callable(env_spec.entry_point) else load_env_creator(env_spec.entry_point)
    env_kwargs = {**copy.deepcopy(env_spec.kwargs), **kwargs}
    env = creator(**env_kwargs)
    made = copy.deepcopy(env_spec)
    made.kwargs = {k: v for k, v in env_kwargs.items() if isinstance(v, (int, float, str, bool, type(None)))}
    env.unwrapped.spec = made
    if env_spec.order_enforce:
        env = wrappers.OrderEnforcing(env)
    steps = max_episode_steps if max_episode_steps is not None else env_spec.max_episode_steps
    if steps is not None:
        env = wrappers.TimeLimit(env, steps)
    return env
''',
}


def write(root):
    """write the package under ``root`` (a directory to be put on sys.path / PYTHONPATH)"""
    import os
    for rel, src in FILES.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(src.lstrip("\n"))
    return root
