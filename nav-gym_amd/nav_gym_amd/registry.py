"""gym registration for NavGym-v0.

The reference registers through gym.envs.registration.register (nav_gym_env/__init__.py:4-40).
gym / gymnasium are not installed in the target image, so this module defers to the real package
when it is importable and otherwise provides the ~40 lines of registry / spaces the drop-in needs:
`register`, `make`, `spaces.Box`, `spaces.Dict`.
"""
import importlib

import numpy as np

try:                                    # the real thing, when present
    import gym as _gym                  # noqa: F401
    from gym import spaces
    from gym.envs.registration import register as _gym_register
    HAVE_GYM = True
except Exception:                       # pragma: no cover - exercised in this image
    _gym = None
    HAVE_GYM = False

    class _Space(object):
        pass

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.dtype = np.dtype(dtype)
            if shape is None:
                low = np.asarray(low, dtype=self.dtype)
                high = np.asarray(high, dtype=self.dtype)
                shape = low.shape
            else:
                low = np.full(shape, low, dtype=self.dtype)
                high = np.full(shape, high, dtype=self.dtype)
            self.low, self.high, self.shape = low, high, tuple(shape)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1e3)
            hi = np.where(np.isfinite(self.high), self.high, 1e3)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    class Dict(_Space):
        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

    class _Spaces(object):
        pass
    spaces = _Spaces()
    spaces.Box, spaces.Dict = Box, Dict

_REGISTRY = {}


def register(id, entry_point, kwargs=None, **extra):
    """Same call shape as gym.envs.registration.register (nav_gym_env/__init__.py:4)."""
    _REGISTRY[id] = dict(entry_point=entry_point, kwargs=dict(kwargs or {}))
    if HAVE_GYM:
        try:
            _gym_register(id=id, entry_point=entry_point, kwargs=kwargs, **extra)
        except Exception:               # already registered
            pass


def spec(id):
    return _REGISTRY[id]


def make(id, **overrides):
    """gym.make('NavGym-v0', **overrides): registered kwargs, caller's overrides on top."""
    ent = _REGISTRY[id]
    mod, cls = ent["entry_point"].split(":")
    kw = dict(ent["kwargs"])
    kw.update(overrides)
    return getattr(importlib.import_module(mod), cls)(**kw)
