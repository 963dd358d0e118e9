"""Per-env drop-in surface: `MultiGridEnv` with the reference's dict-of-agents `reset()` / `step()`.

Mirrors the public interface of `multigrid.base.MultiGridEnv` (multigrid/base.py:36-841) for the hot path:
constructor keywords (base.py:85-103), `reset` (250-301), `step` (303-346), `gen_obs` (348-376), `is_done`
(534-539), `observation_space` / `action_space` (209-227), `agents`, `grid.state`, `agent_states`,
`step_count`, `max_steps`, `unwrapped`.  One instance = one env = a batch-of-1 `BatchedMultiGridEnv`; every
`step` launches the fused HIP kernel and copies the results back, so this class is for drop-in use and
parity checks, not for throughput (use `BatchedMultiGridEnv` for that).

User-defined envs (round 4): a subclass that overrides the reference's extension point `_gen_grid(width, height)`
(base.py:229-247) -- `self.grid = Grid(width, height)`, `self.grid.wall_rect(...)`, `self.put_obj(Goal(), x, y)`,
`self.place_obj(...)`, `self.place_agent(agent)`, `agent.state.pos = ...` -- runs unchanged: `reset()` executes it on the host
against `multigrid_amd.world.Grid` / `WorldObj` value classes, checks the result and uploads it (see `MultiGridEnv.reset`).

Out of scope: `render()` and everything pygame, `place_obj`-style editing of a live
grid (outside `_gen_grid`), user-defined object types.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any

import numpy as np
import torch

from . import layouts, rng as rnglib, world
from .batched import BatchedMultiGridEnv
from .constants import EMPTY_CELL, NO_ACTION, Action, Color, Direction, Type
from .mission import Mission, MissionSpace
from .spaces import Box, Dict, Discrete
from .spec import EnvSpec


class AgentStateRow(np.ndarray):
    """One agent's `(9,)` int row `[type, color, dir, x, y, terminated, carry_type, carry_color, carry_state]` with the attribute
    names of the reference's `AgentState` (multigrid/core/agent.py:222-346).  While `_gen_grid` runs it is a live view of the
    episode's initial agent rows (`agent.state.pos = (1, 1)` places the agent, empty.py:164-167); at any other time it is a
    snapshot of the device-resident row."""

    @property
    def color(self) -> Color:
        return Color(int(self[1]))

    @color.setter
    def color(self, value):
        self[1] = world._index(Color, value, "color")

    @property
    def dir(self) -> int:
        return int(self[2])

    @dir.setter
    def dir(self, value):
        self[2] = int(value)

    @property
    def pos(self) -> tuple[int, int]:
        return (int(self[3]), int(self[4]))

    @pos.setter
    def pos(self, value):
        self[3:5] = (int(value[0]), int(value[1]))

    @property
    def terminated(self) -> bool:
        return bool(self[5])

    @terminated.setter
    def terminated(self, value):
        self[5] = int(bool(value))

    @property
    def carrying(self):
        """The carried object (a `WorldObj`) or None (agent.py:326-346)."""
        return world.WorldObj.from_array(np.asarray(self[6:9]))

    @carrying.setter
    def carrying(self, obj):
        self[6:9] = EMPTY_CELL if obj is None else tuple(obj.encode())


class Agent:
    """One agent, with the constructor and attribute names of `multigrid.core.agent.Agent` (agent.py:22-167).  Built by the
    env for `agents=<int>`, or by the caller and handed over as `agents=[Agent(0), Agent(1), ...]` (base.py:170-177).
    Inside an env its `state` is a read-only view of the env's device-resident agent row."""

    def __init__(self, index: int, mission_space: MissionSpace | str = "maximize reward", view_size: int = 7,
                 see_through_walls: bool = False, *, _env: "MultiGridEnv | None" = None):
        # multigrid/core/agent.py:78-79
        assert view_size % 2 == 1
        assert view_size >= 3
        if isinstance(mission_space, str):
            mission_space = MissionSpace.from_string(mission_space)
        self._env = _env
        self.index = index
        self.view_size = view_size
        self.see_through_walls = see_through_walls
        self.mission: Mission | None = None
        self.observation_space = Dict({                                  # agent.py:85-94
            "image": Box(low=0, high=255, shape=(view_size, view_size, 3), dtype=int),
            "direction": Discrete(len(Direction)),
            "mission": mission_space,
        })
        self.action_space = Discrete(len(Action))                        # agent.py:97

    @property
    def state(self) -> AgentStateRow:
        """(9,) int row: [type, color, dir, x, y, terminated, carry_type, carry_color, carry_state] (`AgentStateRow`: also
        `.pos`, `.dir`, `.color`, `.terminated`, `.carrying` as in multigrid/core/agent.py:222-346)."""
        if self._env is None:                                    # not in an env yet: a fresh AgentState row (agent.py:234-254)
            return layouts._fresh_agents(self.index + 1)[self.index].view(AgentStateRow)
        gen = self._env._gen_agents
        if gen is not None:                                      # inside _gen_grid: the episode's initial rows, live
            return gen[self.index].view(AgentStateRow)
        return self._env.agent_states[self.index].view(AgentStateRow)

    @property
    def color(self) -> Color:
        return Color(int(self.state[1]))

    @property
    def dir(self) -> Direction:
        return Direction(int(self.state[2]))

    @property
    def pos(self) -> tuple[int, int]:
        s = self.state
        return (int(s[3]), int(s[4]))

    @property
    def terminated(self) -> bool:
        return bool(self.state[5])

    @property
    def carrying(self):
        """The carried cell as a (type, color, state) tuple, or None (agent.py:326-346; `agent.state.carrying` gives the
        `WorldObj`)."""
        c = tuple(int(v) for v in self.state[6:9])
        return None if c[0] == Type.empty else c

    @property
    def front_pos(self) -> tuple[int, int]:
        from .constants import DIR_TO_VEC
        dx, dy = DIR_TO_VEC[int(self.state[2])]
        return (int(self.state[3] + dx), int(self.state[4] + dy))

    def encode(self) -> tuple[int, int, int]:
        s = self.state
        return (int(Type.agent), int(s[1]), int(s[2]))


AgentView = Agent          # (round-1 name)


class GridView:
    """`env.grid`: exposes `state` (W,H,3) int64 indexed [x, y] like multigrid/core/grid.py:54."""

    def __init__(self, env: "MultiGridEnv"):
        self._env = env

    @property
    def width(self) -> int:
        return self._env.width

    @property
    def height(self) -> int:
        return self._env.height

    @property
    def state(self) -> np.ndarray:
        return layouts.grid_from_product(self._env._benv.grid[0].cpu().numpy())

    def get(self, x: int, y: int):
        """(type, color, state) at (x, y), or None for an empty cell (grid.py:102-117)."""
        c = tuple(int(v) for v in self.state[x, y])
        return None if c[0] == Type.empty else c

    def encode(self) -> np.ndarray:
        return self.state


class MultiGridEnv:
    """Base class.  A subclass supplies the episode start either the reference's way -- `_gen_grid(width, height)`
    (multigrid/base.py:229-247; see the module docstring) -- or as `_gen_layout` (ready-made product tensors: what the
    built-in env classes do)."""

    metadata = {"render_modes": [], "render_fps": 20}
    env_kind = "empty"

    def __init__(
            self,
            mission_space: MissionSpace | str = "maximize reward",
            agents: int = 1,
            grid_size: int | None = None,
            width: int | None = None,
            height: int | None = None,
            max_steps: int = 100,
            see_through_walls: bool = False,
            agent_view_size: int = 7,
            allow_agent_overlap: bool = True,
            joint_reward: bool = False,
            success_termination_mode: str = "any",
            failure_termination_mode: str = "all",
            render_mode: str | None = None,
            device="cuda",
            layout_seed: int | None = None,
            _backend=None,
            **unused_render_kwargs: Any):
        """Keyword arguments as multigrid/base.py:85-103.  Extras: `device` (HIP device), `layout_seed` (seed of
        the construction-time generator that drives object placement -- the reference takes it from OS
        entropy)."""
        if render_mode is not None:
            raise NotImplementedError("rendering is out of scope for multigrid_amd (SURVEY.md section 2)")
        self._gen_agents: np.ndarray | None = None        # (A,9) initial agent rows while a user's _gen_grid runs
        given_agents = None
        if not isinstance(agents, int):                                       # base.py:170-177: an iterable of Agent objects
            try:
                given_agents = sorted(agents, key=lambda agent: agent.index)
            except (TypeError, AttributeError):
                raise ValueError(f"Invalid argument for agents: {agents}")
            assert {agent.index for agent in given_agents} == set(range(len(given_agents)))
            agents = len(given_agents)
            # gen_obs renders every agent with agents[0]'s view (base.py:364-365)
            agent_view_size, see_through_walls = given_agents[0].view_size, given_agents[0].see_through_walls
        self.mission_space = (MissionSpace.from_string(mission_space) if isinstance(mission_space, str)
                              else mission_space)
        width, height = (grid_size, grid_size) if grid_size else (width, height)
        assert width is not None and height is not None
        self.width, self.height = width, height
        assert isinstance(max_steps, int), f"The argument max_steps must be an integer, got: {type(max_steps)}"
        self.spec = EnvSpec(
            width=width, height=height, num_agents=agents, view_size=agent_view_size, max_steps=max_steps,
            see_through_walls=see_through_walls, allow_agent_overlap=allow_agent_overlap,
            joint_reward=joint_reward, success_termination_mode=success_termination_mode,
            failure_termination_mode=failure_termination_mode, env_kind=self.env_kind)
        self.num_agents = agents
        self.max_steps = max_steps
        self.allow_agent_overlap = allow_agent_overlap
        self.joint_reward = joint_reward
        self.success_termination_mode = success_termination_mode
        self.failure_termination_mode = failure_termination_mode
        self.render_mode = None
        self.actions = Action
        self.reward_range = (0, 1)
        if given_agents is None:
            self.agents = [Agent(i, self.mission_space, agent_view_size, see_through_walls, _env=self)
                           for i in range(agents)]
        else:
            self.agents = given_agents
            for agent in self.agents:
                agent._env = self                                             # agent.state now views the joint state
        self.grid = GridView(self)
        self.mission: Mission | str | None = None
        if callable(_backend):
            _backend = _backend(self.spec)          # test-suite hook: factory(spec) -> launcher
        self._benv = BatchedMultiGridEnv(self.spec, 1, device, backend=_backend)
        # construction-time generator: every `_rand_*` placement draw (multigrid/base.py:143, utils/random.py:14-21)
        self._layout_rng = rnglib.seeded_generator(layout_seed)
        self._np_random: np.random.Generator | None = None
        self._rng_on_device = False

    # ------------------------------------------------------------------------------------ gym.Env surface
    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self) -> np.random.Generator:
        """`env.np_random` (the seeded stream the action order is drawn from).  While an episode is running
        the live state is on the device; this returns a host generator at that state."""
        self._pull_rng()
        return self._np_random

    def _pull_rng(self):
        if self._np_random is None:
            self._np_random = rnglib.seeded_generator(None)
        elif self._rng_on_device and self.num_agents > 1:
            # Only the 128-bit LCG state moved on the device (Generator.random() never touches numpy's buffered
            # uint32); keep the host generator object so that `has_uint32` / `uinteger` left behind by an earlier
            # `integers()` draw (RoomGrid door placement) carry over exactly as in the reference.
            words = [int(w) for w in self._benv.rng[0].cpu().numpy().view(np.uint64)]
            st = self._np_random.bit_generator.state
            st["state"]["state"] = words[0] | (words[1] << 64)
            self._np_random.bit_generator.state = st
        self._rng_on_device = False

    @property
    def observation_space(self):
        return Dict({agent.index: agent.observation_space for agent in self.agents})   # base.py:209-217

    @property
    def action_space(self):
        return Dict({agent.index: agent.action_space for agent in self.agents})        # base.py:219-227

    @property
    def step_count(self) -> int:
        return int(self._benv.step_count[0])

    @property
    def agent_states(self) -> np.ndarray:
        """(A,9) int64 snapshot in the reference's AgentState column order (agent.py:222-232)."""
        return layouts.unpack_agents(self._benv.agents[0].cpu().numpy())

    def _gen_layout(self, layout_rng: np.random.Generator, np_random: np.random.Generator):
        """Return (grid u8[H,W,3], agents u8[A,8], aux u8[16] | None) for a new episode.  The default runs the subclass'
        `_gen_grid` (the reference's extension point) on the host and converts what it built."""
        if type(self)._gen_grid is MultiGridEnv._gen_grid:
            raise NotImplementedError(f"{type(self).__name__} defines neither _gen_grid(width, height) nor _gen_layout()")
        if self.env_kind != "empty":
            raise NotImplementedError("a user-defined _gen_grid is supported for hook-free envs (env_kind 'empty'): the built-in "
                                      "hook envs carry per-episode hook state that their own generators fill in")
        live_grid = self.grid
        self._gen_agents = layouts._fresh_agents(self.num_agents)             # base.py:275-277: AgentState(num_agents) + reset
        try:
            self._gen_grid(self.width, self.height)                           # base.py:280
            host = self.grid
            if not isinstance(host, world.Grid):
                raise TypeError("_gen_grid must set self.grid = Grid(width, height) (multigrid/base.py:229-247)")
            if (host.width, host.height) != (self.width, self.height):
                raise ValueError(f"_gen_grid built a {host.width}x{host.height} grid for a {self.width}x{self.height} env")
            ag9 = self._gen_agents
            # base.py:283-284
            assert np.all(ag9[:, 3:5] >= 0)
            assert np.all(ag9[:, 2] >= 0)
            grid = layouts.grid_to_product(host.state)
            layouts.check_walled(grid)                                        # the kernels' precondition (include/mgx.h)
            return grid, layouts.pack_agents(ag9), None
        finally:
            self._gen_agents = None
            self.grid = live_grid                                             # `env.grid` shows the device-resident state again

    # ------------------------------------------------------------------------------------ the reference's extension point
    def _gen_grid(self, width: int, height: int):
        """multigrid/base.py:229-247: generate the grid for a new episode -- set `self.grid` and populate it with `WorldObj`s,
        set the position and direction of every agent.  Override it in a subclass exactly as with the reference."""
        raise NotImplementedError

    def put_obj(self, obj, i: int, j: int):
        """multigrid/base.py:659-665"""
        self.grid.set(i, j, obj)
        obj.init_pos = (i, j)
        obj.cur_pos = (i, j)

    def place_obj(self, obj, top=None, size=None, reject_fn=None, max_tries=float("inf")):
        """multigrid/base.py:604-657: rejection sampling of an empty cell in the rectangle (top, size); the draws come from the
        construction-time generator (`_rand_int`), as in the reference."""
        if self._gen_agents is None:
            raise RuntimeError("place_obj edits the episode being generated: it is available inside _gen_grid only")
        top = (0, 0) if top is None else (max(top[0], 0), max(top[1], 0))
        if size is None:
            size = (self.grid.width, self.grid.height)
        num_tries = 0
        while True:
            if num_tries > max_tries:
                raise RecursionError("rejection sampling failed in place_obj")
            num_tries += 1
            pos = (self._rand_int(top[0], min(top[0] + size[0], self.grid.width)),
                   self._rand_int(top[1], min(top[1] + size[1], self.grid.height)))
            if self.grid.get(*pos) is not None:                               # not on top of another object
                continue
            if ((self._gen_agents[:, 3] == pos[0]) & (self._gen_agents[:, 4] == pos[1])).any():     # not where agents are
                continue
            if reject_fn and reject_fn(self, pos):
                continue
            break
        self.grid.set(pos[0], pos[1], obj)
        if obj is not None:
            obj.init_pos = pos
            obj.cur_pos = pos
        return pos

    def place_agent(self, agent, top=None, size=None, rand_dir=True, max_tries=float("inf")):
        """multigrid/base.py:667-686"""
        agent.state.pos = (-1, -1)
        pos = self.place_obj(None, top, size, max_tries=max_tries)
        agent.state.pos = pos
        if rand_dir:
            agent.state.dir = self._rand_int(0, 4)
        return pos

    # multigrid/utils/random.py:9-103 (RandomMixin over the construction-time generator)
    def _rand_int(self, low: int, high: int) -> int:
        return self._layout_rng.integers(low, high)

    def _rand_float(self, low: float, high: float) -> float:
        return self._layout_rng.uniform(low, high)

    def _rand_bool(self) -> bool:
        return self._layout_rng.integers(0, 2) == 0

    def _rand_elem(self, iterable):
        lst = list(iterable)
        return lst[self._rand_int(0, len(lst))]

    def _rand_subset(self, iterable, num_elems: int) -> list:
        lst = list(iterable)
        assert num_elems <= len(lst)
        out = []
        while len(out) < num_elems:
            elem = self._rand_elem(lst)
            lst.remove(elem)
            out.append(elem)
        return out

    def _rand_perm(self, iterable) -> list:
        lst = list(iterable)
        self._layout_rng.shuffle(lst)
        return lst

    def _rand_color(self) -> Color:
        return self._rand_elem(Color)

    def _rand_pos(self, x_low: int, x_high: int, y_low: int, y_high: int) -> tuple[int, int]:
        return (self._layout_rng.integers(x_low, x_high), self._layout_rng.integers(y_low, y_high))

    def reset(self, seed: int | None = None, **kwargs):
        """multigrid/base.py:250-301.  Returns (observations, infos)."""
        if seed is not None:
            self._np_random = rnglib.seeded_generator(seed)          # gym.Env.reset(seed=seed), base.py:269
            self._rng_on_device = False
        else:
            self._pull_rng()
        self.mission_space.seed(seed)                                 # base.py:272-273
        self.mission = self.mission_space.sample()
        for agent in self.agents:
            agent.mission = self.mission                              # base.py:274-277
        grid, agents, aux = self._gen_layout(self._layout_rng, self._np_random)      # base.py:280
        # base.py:283-289: agents placed, not on top of a non-overlappable object
        ag9 = layouts.unpack_agents(agents)
        assert np.all(ag9[:, 3:5] >= 0) and np.all(ag9[:, 2] >= 0)
        for a in ag9:
            t, s = grid[a[4], a[3], 0], grid[a[4], a[3], 2]
            assert t in (Type.empty, Type.goal, Type.floor, Type.lava) or (t == Type.door and s == 0)
        words = rnglib.words_from_bitgen_state(self._np_random.bit_generator.state)
        self._benv.load_state(grid, agents, rng=words, aux=aux)             # step_count = 0 (base.py:292)
        self._rng_on_device = True
        obs, dirs = self._benv.gen_obs()                                     # base.py:295
        return self._obs_dict(obs[0].cpu().numpy(), dirs[0].cpu().numpy()), defaultdict(dict)

    def step(self, actions: dict[int, int]):
        """multigrid/base.py:303-346.  Returns (observations, rewards, terminations, truncations, infos)."""
        A = self.num_agents
        act = np.full((1, A), NO_ACTION, dtype=np.int8)
        keys = []
        for i, a in actions.items():
            if isinstance(i, (int, np.integer)) and 0 <= i < A:          # other keys are never visited (base.py:402-404)
                a = int(a)
                act[0, i] = a if 0 <= a <= 127 else 127                   # out of range -> "unknown action" on device
                keys.append(int(i))
        benv = self._benv
        hook_order = None
        if self.env_kind in ("redbluedoors", "lockedhallway") and keys != sorted(keys):
            # the subclass hooks iterate `actions.items()` (redbluedoors.py:176, locked_hallway.py:210): the dict's insertion
            # order decides who is visited first when two agents toggle the same door in one step
            order = keys + [i for i in range(A) if i not in keys]        # (absent agents carry NO_ACTION: skipped anyway)
            hook_order = torch.tensor([order], dtype=torch.uint8).to(benv.device)
        obs, dirs, rew, term, trunc = benv.step(torch.from_numpy(act).to(benv.device), hook_order=hook_order)
        try:
            benv.check_errors()
        except ValueError:
            bad = [int(a) for a in actions.values() if not 0 <= int(a) <= int(Action.done)]
            raise ValueError(f"Unknown action: {bad[0] if bad else '?'}") from None   # base.py:473-474
        obs, dirs = obs[0].cpu().numpy(), dirs[0].cpu().numpy()
        rew, term = rew[0].cpu().numpy(), term[0].cpu().numpy()
        truncated = bool(trunc[0])
        rewards = {i: (float(rew[i]) if rew[i] != 0 else 0) for i in range(A)}    # int 0 unless rewarded (base.py:393)
        terminations = {i: bool(term[i]) for i in range(A)}
        truncations = {i: truncated for i in range(A)}
        return self._obs_dict(obs, dirs), rewards, terminations, truncations, defaultdict(dict)

    def gen_obs(self):
        """multigrid/base.py:348-376"""
        obs, dirs = self._benv.gen_obs()
        return self._obs_dict(obs[0].cpu().numpy(), dirs[0].cpu().numpy())

    def _obs_dict(self, image_u8: np.ndarray, dirs: np.ndarray):
        image = image_u8.astype(np.int64)                               # the reference's images are `int`
        return {i: {"image": image[i], "direction": int(dirs[i]), "mission": self.agents[i].mission}
                for i in range(self.num_agents)}

    def is_done(self) -> bool:
        """multigrid/base.py:534-539"""
        return bool(self._benv.is_done()[0])

    def render(self):
        raise NotImplementedError("rendering is out of scope for multigrid_amd (SURVEY.md section 2)")

    def close(self):
        pass
