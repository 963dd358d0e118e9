"""Observation / action space descriptors.

The reference builds `gymnasium.spaces` objects (multigrid/core/agent.py:85-97, multigrid/base.py:209-227).
gymnasium is an optional dependency here: when importable its classes are used, otherwise the minimal
stand-ins below provide the attributes callers read (`shape`, `dtype`, `low`, `high`, `n`, dict access).
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is absent in the build image
    from gymnasium.spaces import Box, Dict, Discrete, MultiDiscrete  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class _Space:
        def __init__(self, shape=None, dtype=None):
            self.shape, self.dtype = shape, dtype
            self._np_random = None

        def seed(self, seed=None):
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
            return [seed]

        @property
        def np_random(self):
            if self._np_random is None:
                self.seed()
            return self._np_random

    class Box(_Space):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            super().__init__(tuple(shape) if shape is not None else None, np.dtype(dtype))
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool((x >= self.low).all() and (x <= self.high).all())

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"

    class Discrete(_Space):
        def __init__(self, n, start=0):
            super().__init__((), np.dtype(np.int64))
            self.n, self.start = int(n), int(start)

        def sample(self):
            return int(self.start + self.np_random.integers(self.n))

        def contains(self, x):
            return self.start <= int(x) < self.start + self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiDiscrete(_Space):
        def __init__(self, nvec, dtype=np.int64):
            self.nvec = np.array(nvec, dtype=dtype, copy=True)
            super().__init__(self.nvec.shape, np.dtype(dtype))

        def sample(self):
            # gymnasium.spaces.MultiDiscrete.sample
            return (self.np_random.random(self.nvec.shape) * self.nvec).astype(self.dtype)

    class Dict(_Space, dict):
        def __init__(self, spaces=None, **kw):
            dict.__init__(self, spaces or {}, **kw)
            _Space.__init__(self)

        @property
        def spaces(self):
            return self

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {v!r}" for k, v in self.items()) + ")"
