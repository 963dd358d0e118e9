"""RLlib `MultiAgentEnv`-style adapter (multigrid/rllib/__init__.py:44-105 of the reference).

ray is not a dependency: the wrapper is duck-typed (it subclasses `ray.rllib.env.MultiAgentEnv` only when ray
is importable).  What it adds on top of the env is exactly what the reference adds: the `'__all__'` keys of
`terminations` / `truncations` (rllib/__init__.py:61-62), `agents` / `possible_agents` id lists, and
`get_observation_space` / `get_action_space` (65-69).
"""
from __future__ import annotations

try:  # pragma: no cover
    from ray.rllib.env.multi_agent_env import MultiAgentEnv as _Base  # type: ignore
except Exception:  # noqa: BLE001
    _Base = object


class RLlibWrapper(_Base):
    def __init__(self, env):
        if _Base is not object:  # pragma: no cover
            super().__init__()
        self.env = env
        self.agents = list(range(env.num_agents))
        self.possible_agents = list(range(env.num_agents))
        self.observation_space = env.observation_space
        self.action_space = env.action_space

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, *, seed=None, options=None):
        return self.env.reset(seed=seed)

    def step(self, action_dict):
        obs, rewards, terminations, truncations, infos = self.env.step(action_dict)
        terminations["__all__"] = all(terminations.values())      # rllib/__init__.py:61
        truncations["__all__"] = all(truncations.values())        # rllib/__init__.py:62
        return obs, rewards, terminations, truncations, infos

    def get_observation_space(self, agent_id):
        return self.env.agents[agent_id].observation_space

    def get_action_space(self, agent_id):
        return self.env.agents[agent_id].action_space


def to_rllib_env(env_cls, *wrappers, default_config=None):
    """multigrid/rllib/__init__.py:72-105: a class taking one `config` dict, as RLlib expects."""
    default_config = dict(default_config or {})

    class RLlibEnv(RLlibWrapper):
        def __init__(self, config=None):
            env = env_cls(**{**default_config, **(config or {})})
            for wrapper in wrappers:
                env = wrapper(env)
            super().__init__(env)

    RLlibEnv.__name__ = f"RLlib_{env_cls.__name__}"
    return RLlibEnv
