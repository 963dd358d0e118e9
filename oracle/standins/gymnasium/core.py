from typing import Any, TypeVar

import numpy as np

ActType = TypeVar("ActType")
ObsType = TypeVar("ObsType")


def _np_random(seed=None):
    # gymnasium.utils.seeding.np_random: Generator(PCG64(SeedSequence(seed)))
    seed_seq = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(seed_seq)), seed_seq.entropy


class Env:
    metadata: dict = {"render_modes": []}
    render_mode = None
    _np_random = None
    # gymnasium seeds the construction-time generator from OS entropy; gen_golden.py pins it so that
    # regenerating the fixtures is reproducible (it only drives layout placement).
    _default_seed = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, _ = _np_random(type(self)._default_seed)
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, _ = _np_random(seed)

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed, options=options)

    def step(self, action):
        return self.env.step(action)


class ObservationWrapper(Wrapper):
    def reset(self, *, seed=None, options=None):
        obs, info = self.env.reset(seed=seed)
        return self.observation(obs), info

    def step(self, action):
        obs, reward, terminated, truncated, info = self.env.step(action)
        return self.observation(obs), reward, terminated, truncated, info

    def observation(self, observation):
        raise NotImplementedError
