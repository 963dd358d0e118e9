"""Observation wrappers with the reference's names and behaviour (multigrid/wrappers.py), over the dict API.

The heavy lifting is done by the HIP kernels behind `BatchedMultiGridEnv.one_hot_obs()` / `.full_obs()`; these
classes only reshape their results into the per-agent dicts the reference returns.
"""
from __future__ import annotations

import numpy as np

from .spaces import Box


class _ObservationWrapper:
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, seed=None, **kwargs):
        obs, info = self.env.reset(seed=seed, **kwargs)
        return self.observation(obs), info

    def step(self, actions):
        obs, reward, terminated, truncated, info = self.env.step(actions)
        return self.observation(obs), reward, terminated, truncated, info

    def observation(self, obs):
        raise NotImplementedError


class FullyObsWrapper(_ObservationWrapper):
    """multigrid/wrappers.py:17-58: every agent's 'image' is the whole grid (W,H,3) with all agents drawn in."""

    def __init__(self, env):
        super().__init__(env)
        for agent in self.env.unwrapped.agents:
            agent.observation_space["image"] = Box(low=0, high=255, shape=(env.height, env.width, 3), dtype=int)

    def observation(self, obs):
        img = self.env.unwrapped._benv.full_obs()[0].cpu().numpy().astype(np.int64)
        for agent_id in obs:
            obs[agent_id]["image"] = img            # one shared array, as in the reference
        return obs


class ImgObsWrapper(_ObservationWrapper):
    """multigrid/wrappers.py:61-98: keep only the image."""

    def __init__(self, env):
        super().__init__(env)
        for agent in self.env.unwrapped.agents:
            agent.observation_space = agent.observation_space["image"]
            agent.observation_space.dtype = np.uint8                 # wrappers.py:88-89

    def observation(self, obs):
        return {agent_id: o["image"].astype(np.uint8) for agent_id, o in obs.items()}          # wrappers.py:95-96


class OneHotObsWrapper(_ObservationWrapper):
    """multigrid/wrappers.py:101-190: (v,v,3) int -> (v,v,21) uint8 one-hot of (type, color, state/direction)."""

    def __init__(self, env):
        super().__init__(env)
        self.dim_sizes = np.array([11, 6, 4])       # len(Type), len(Color), max(len(State), len(Direction))
        dim = int(self.dim_sizes.sum())
        for agent in self.env.unwrapped.agents:
            h, w, _ = agent.observation_space["image"].shape
            agent.observation_space["image"] = Box(low=0, high=1, shape=(h, w, dim), dtype=np.uint8)

    def observation(self, obs):
        base = self.env.unwrapped
        if self.env is base:
            # directly over the env: its partial views are still in HBM, one-hot them there
            oh = base._benv.one_hot_obs()[0].cpu().numpy()
            for agent_id in obs:
                obs[agent_id]["image"] = oh[agent_id]
            return obs
        # over another wrapper (e.g. FullyObsWrapper): one-hot the images handed in, as wrappers.py:149-156 does
        import torch
        benv = base._benv
        for agent_id in list(obs):
            img = torch.from_numpy(np.ascontiguousarray(obs[agent_id]["image"]).astype(np.uint8)).to(benv.device)
            out = torch.empty(tuple(img.shape[:-1]) + (int(self.dim_sizes.sum()),), dtype=torch.uint8, device=benv.device)
            benv.backend.one_hot(img, out)
            obs[agent_id]["image"] = out.cpu().numpy()
        return obs


class SingleAgentWrapper:
    """multigrid/wrappers.py:193-233: single-agent view of a one-agent env (plain obs / action instead of dicts)."""

    def __init__(self, env):
        self.env = env
        self.observation_space = env.agents[0].observation_space
        self.action_space = env.agents[0].action_space

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, *args, **kwargs):
        result = self.env.reset(*args, **kwargs)
        return tuple(item[0] for item in result)

    def step(self, action):
        result = self.env.step({0: action})
        return tuple(item[0] for item in result)
