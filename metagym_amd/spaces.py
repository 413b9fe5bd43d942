"""Minimal gym.spaces look-alikes (gym itself is not a dependency of this engine).

Only what the reference envs declare: Box, Discrete and the bare Space used by
metagym/quadrotor/env.py:96. `sample()` returns a batch when `num_envs` is given.
"""
import numpy as np


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self.np_random = np.random.RandomState()

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(seed)
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is None:
            shape = low.shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(low, self.shape).copy()
        self.high = np.broadcast_to(high, self.shape).copy()

    def sample(self, num_envs=None):
        shape = self.shape if num_envs is None else (num_envs,) + self.shape
        return self.np_random.uniform(self.low, self.high, size=shape).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape[-len(self.shape):] == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = int(n)

    def sample(self, num_envs=None):
        if num_envs is None:
            return int(self.np_random.randint(self.n))
        return self.np_random.randint(self.n, size=(num_envs,)).astype(np.int64)

    def contains(self, x):
        x = np.asarray(x)
        return bool(np.all(x >= 0) and np.all(x < self.n))
