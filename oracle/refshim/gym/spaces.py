"""gym.spaces stand-in (see gym/__init__.py docstring)."""
from typing import Generic, TypeVar

import numpy as np

from .utils import seeding

T_cov = TypeVar("T_cov", covariant=True)


class Space(Generic[T_cov]):
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._np_random = None
        if seed is not None:
            self._np_random, _ = seeding.np_random(seed)

    @property
    def shape(self):
        return self._shape

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, _ = seeding.np_random()
        return self._np_random

    def sample(self):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        super().__init__(shape, dtype, seed)
        self.low = np.full(self._shape, low, dtype=self.dtype)
        self.high = np.full(self._shape, high, dtype=self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self._shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64, seed)
        self.n = int(n)
        self.start = int(start)

    def sample(self):
        return int(self.start + self.np_random.integers(self.n))

    def contains(self, x):
        return self.start <= int(x) < self.start + self.n


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype, seed)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.nvec.shape and bool(np.all(x >= 0) and np.all(x < self.nvec))


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        super().__init__(None, None, seed)
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def contains(self, x):
        return isinstance(x, dict) and all(k in x for k in self.spaces)
