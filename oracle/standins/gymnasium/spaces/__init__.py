import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self.shape = shape
        self.dtype = dtype
        self._rng = None

    @property
    def np_random(self):
        if self._rng is None:
            self.seed()
        return self._rng

    def seed(self, seed=None):
        self._rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        return [seed]


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        super().__init__(shape, dtype, seed)
        self.low, self.high = low, high


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64, seed)
        self.n, self.start = int(n), int(start)

    def sample(self):
        return int(self.start + self.np_random.integers(self.n))


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.array(nvec, dtype=dtype, copy=True)
        super().__init__(self.nvec.shape, dtype, seed)

    def sample(self):
        return (self.np_random.random(self.nvec.shape) * self.nvec).astype(self.dtype)


class Dict(Space, dict):
    def __init__(self, spaces=None, seed=None, **kw):
        dict.__init__(self, spaces or {}, **kw)
        Space.__init__(self, None, None, seed)

    @property
    def spaces(self):
        return self
