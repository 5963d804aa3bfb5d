"""Gymnasium surface used by the env adapters.

If ``gymnasium`` is importable it is used as-is (so ``gym.make('ship-ice-v0', cfg=...)`` goes through the real
registry and TimeLimit wrapper, like benchpush/environments/__init__.py:3-7).  This image ships without it, so a
minimal stand-in with the same names (``Env``, ``spaces.Box``, ``register``, ``make`` and a ``TimeLimit`` wrapper
that honours ``max_episode_steps`` and ``.unwrapped``) is provided.
"""
import importlib

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import gymnasium as _gym
    from gymnasium import spaces  # noqa: F401
    Env = _gym.Env
    register = _gym.register
    make = _gym.make
    HAVE_GYMNASIUM = True
except ImportError:
    HAVE_GYMNASIUM = False

    class _Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low)
            self.shape = tuple(shape)
            self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, self.dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            if np.issubdtype(self.dtype, np.integer):
                return self._rng.integers(self.low, self.high, size=self.shape, endpoint=True).astype(self.dtype)
            return self._rng.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)

    class spaces:  # noqa: N801 - mirrors `from gymnasium import spaces`
        Box = _Box

    class Env:
        metadata = {}
        action_space = None
        observation_space = None

        @property
        def unwrapped(self):
            return self

        def reset(self, seed=None, options=None):
            raise NotImplementedError

        def step(self, action):
            raise NotImplementedError

        def render(self):
            raise NotImplementedError

        def close(self):
            pass

    class TimeLimit(Env):
        def __init__(self, env, max_episode_steps):
            self.env = env
            self._max = max_episode_steps
            self._t = 0
            self.action_space = env.action_space
            self.observation_space = env.observation_space

        @property
        def unwrapped(self):
            return self.env.unwrapped

        def __getattr__(self, name):
            return getattr(self.env, name)

        def reset(self, **kw):
            self._t = 0
            return self.env.reset(**kw)

        def step(self, action):
            obs, r, term, trunc, info = self.env.step(action)
            self._t += 1
            if self._t >= self._max:
                trunc = True
            return obs, r, term, trunc, info

        def render(self):
            return self.env.render()

        def close(self):
            return self.env.close()

    _REGISTRY = {}

    def register(id, entry_point, max_episode_steps=None, **kwargs):  # noqa: A002
        _REGISTRY[id] = (entry_point, max_episode_steps, kwargs)

    def make(id, **kwargs):  # noqa: A002
        if id not in _REGISTRY:
            raise KeyError("unknown environment id %r" % (id,))
        entry_point, max_steps, base_kw = _REGISTRY[id]
        if isinstance(entry_point, str):
            mod, cls = entry_point.split(":")
            entry_point = getattr(importlib.import_module(mod), cls)
        env = entry_point(**{**base_kw, **kwargs})
        if max_steps is not None:
            env = TimeLimit(env, max_steps)
        return env
