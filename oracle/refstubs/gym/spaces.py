import numpy as np


class Space(object):
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        self._rng = np.random.RandomState(0)

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        low = np.asarray(low)
        high = np.asarray(high)
        if shape is None:
            shape = low.shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(low, self.shape).astype(self.dtype)
        self.high = np.broadcast_to(high, self.shape).astype(self.dtype)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)


class Discrete(Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = n

    def sample(self):
        return int(self._rng.randint(self.n))


class Dict(Space):
    """gym.spaces.Dict: an ordered mapping of named sub-spaces (quadrupedal/envs/sensors/space_utils.py:115)."""
    def __init__(self, spaces=None):
        super().__init__(None, None)
        import collections
        self.spaces = collections.OrderedDict(spaces or {})

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}
