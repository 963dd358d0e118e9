"""Per-env drop-in surface: `MultiGridEnv` with the reference's dict-of-agents `reset()` / `step()`.

Mirrors the public interface of `multigrid.base.MultiGridEnv` (multigrid/base.py:36-841) for the hot path:
constructor keywords (base.py:85-103), `reset` (250-301), `step` (303-346), `gen_obs` (348-376), `on_success` / `on_failure`
(478-532), `is_done` (534-539), `observation_space` / `action_space` (209-227), `agents`, `grid`, `agent_states`,
`step_count`, `max_steps`, `unwrapped`.  One instance = one env = a batch-of-1 `BatchedMultiGridEnv`; every
`step` launches the fused HIP kernel and copies the results back, so this class is for drop-in use and
parity checks, not for throughput (use `BatchedMultiGridEnv` for that).

User-defined envs.  Both of the reference's extension points run unchanged:
  * `_gen_grid(width, height)` (base.py:229-247) -- `self.grid = Grid(width, height)`, `self.grid.wall_rect(...)`,
    `self.put_obj(Goal(), x, y)`, `self.place_obj(...)`, `self.place_agent(agent)`, `agent.pos = ...` -- is executed by `reset()` on
    the host against `multigrid_amd.world.Grid` / `WorldObj`, its result checked and uploaded (`MultiGridEnv._gen_layout`);
  * a `step(actions)` override the way the reference's own envs end their episodes (envs/blockedunlockpickup.py:166-175,
    envs/redbluedoors.py:170-187): `obs, reward, terminated, truncated, info = super().step(actions)`, then plain Python over the
    post-step state -- `agent.state.carrying == self.obj`, `self.grid.get(*agent.front_pos) == self.door`, `self.door.is_open` --
    and `self.on_success(agent, reward, terminated)` / `self.on_failure(...)`.  The base step is the kernel; what the hook reads is
    pulled from the device, what it changes (terminated flags, a door's state through `grid.update`, `grid.set`) is written back
    before the next step.  The objects a `_gen_grid` placed keep their identity: `grid.get` / `agent.state.carrying` hand back the
    very object that was put there (matched by what the device cell holds), so the reference's `==` (identity,
    world_object.py:126-127) means the same here.

Out of scope: `render()` and everything pygame, user-defined object types.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any

import numpy as np
import torch

from . import layouts, rng as rnglib, world
from .batched import BatchedMultiGridEnv
from .constants import DIR_TO_VEC, EMPTY_CELL, NO_ACTION, Action, Color, Direction, Type
from .mission import Mission, MissionSpace
from .spaces import Box, Dict, Discrete
from .spec import EnvSpec


class AgentStateRow(np.ndarray):
    """Agent state in the reference's `AgentState` layout (multigrid/core/agent.py:170-346): rows of `dim` = 9 ints `[type, color,
    dir, x, y, terminated, carry_type, carry_color, carry_state]` with the attribute names and index constants of the reference.
    One row (`agent.state`) or all of an env's rows (`env.agent_states`).

    While `_gen_grid` runs, a row is a live view of the episode's initial agent rows (`agent.state.pos = (1, 1)` places the agent,
    empty.py:164-167).  At any other time it is a snapshot of the device-resident rows whose SETTERS write through to the device
    (`agent.state.terminated = True`, `env.agent_states.terminated = True`: what `on_success` does, base.py:494-498); reading after
    a `step` needs a fresh `agent.state`."""

    # State vector indices (agent.py:222-232)
    TYPE = 0
    COLOR = 1
    DIR = 2
    ENCODING = slice(0, 3)
    POS = slice(3, 5)
    TERMINATED = 5
    CARRYING = slice(6, 9)
    dim = 9

    def __array_finalize__(self, obj):
        self._env = getattr(obj, "_env", None)           # the MultiGridEnv whose device state a setter writes through to
        self._index = getattr(obj, "_index", None)       # the agent this row is (None: all agents)
        self._content = getattr(obj, "_content", 0)      # what a carried box holds (include/mgx.h "BOX CONTENTS"): not part of the
        self._contents = getattr(obj, "_contents", None)  # row; per agent for the (A,9) form

    def __getitem__(self, idx):
        out = super().__getitem__(idx)
        if self.ndim == 2 and isinstance(idx, (int, np.integer)) and isinstance(out, AgentStateRow):
            out._index = int(idx)                        # one agent's row of `env.agent_states`
            out._content = int(self._contents[idx]) if self._contents is not None else 0
        return out

    # -- write-through
    def _push(self, cols, value):
        self[..., cols] = value
        env = self._env
        if env is not None and env._gen_agents is None:
            env._write_agents(self._index, cols, value)

    @property
    def color(self):
        v = self[..., 1]
        return Color(int(v)) if v.ndim == 0 else np.asarray(v)

    @color.setter
    def color(self, value):
        self._push(1, world._index(Color, value, "color"))

    @property
    def dir(self):
        v = self[..., 2]
        return int(v) if v.ndim == 0 else np.asarray(v)

    @dir.setter
    def dir(self, value):
        self._push(2, np.asarray(value, dtype=np.int64))

    @property
    def pos(self):
        v = self[..., 3:5]
        return (int(v[0]), int(v[1])) if v.ndim == 1 else np.asarray(v)

    @pos.setter
    def pos(self, value):
        self._push(slice(3, 5), np.asarray(value, dtype=np.int64))

    @property
    def terminated(self):
        v = self[..., 5]
        return bool(v) if v.ndim == 0 else np.asarray(v).astype(bool)

    @terminated.setter
    def terminated(self, value):
        self._push(5, np.asarray(value).astype(np.int64))

    @property
    def carrying(self):
        """The carried object (a `WorldObj`) or None (agent.py:326-346).  Inside an env it is the very object the layout placed
        when the carried cell identifies one (`agent.state.carrying == self.obj`, blockedunlockpickup.py:172)."""
        if self.ndim != 1:
            raise AttributeError("carrying: one agent's row at a time")
        cell = [int(v) for v in self[6:9]]
        if self._content:
            cell[2] |= int(self._content) << 2
        env = self._env
        if env is not None and env._gen_agents is None:
            return env._object_for(cell, pos=None, agent=self._index)
        return world.WorldObj.from_array(cell)

    @carrying.setter
    def carrying(self, obj):
        if self.ndim != 1:
            raise AttributeError("carrying: one agent's row at a time")
        cell = list(EMPTY_CELL) if obj is None else list(obj.encode())
        self._content = world.content_code(getattr(obj, "contains", None)) if obj is not None else 0
        self[6:9] = cell
        env = self._env
        if env is not None and env._gen_agents is None:
            cell[2] |= self._content << 2
            env._write_agents(self._index, slice(6, 9), np.asarray(cell))
            if obj is not None:
                env._adopt(obj, None)
            env._bind_carried(self._index, obj)
        elif env is not None and self._content:
            env._gen_carry_content[self._index] = self._content


class Agent:
    """One agent, with the constructor and attribute names of `multigrid.core.agent.Agent` (agent.py:22-167).  Built by the
    env for `agents=<int>`, or by the caller and handed over as `agents=[Agent(0), Agent(1), ...]` (base.py:170-177).
    `color` / `dir` / `pos` / `terminated` / `carrying` alias `state` both ways, as the reference's `PropertyAlias` does
    (agent.py:100-109)."""

    def __init__(self, index: int, mission_space: MissionSpace | str = "maximize reward", view_size: int = 7,
                 see_through_walls: bool = False, *, _env: "MultiGridEnv | None" = None):
        # multigrid/core/agent.py:78-79
        assert view_size % 2 == 1
        assert view_size >= 3
        if isinstance(mission_space, str):
            mission_space = MissionSpace.from_string(mission_space)
        self._env = _env
        self._own = None                                                 # its AgentState row while it belongs to no env
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
        """This agent's `AgentState` row (see `AgentStateRow`)."""
        env = self._env
        if env is None:                                          # not in an env yet: its own AgentState row (agent.py:234-254)
            if self._own is None:
                self._own = layouts._fresh_agents(self.index + 1)[self.index].copy().view(AgentStateRow)
            return self._own
        gen = env._gen_agents
        if gen is not None:                                      # inside _gen_grid: the episode's initial rows, live
            row = gen[self.index].view(AgentStateRow)
            row._env, row._index = env, self.index
            return row
        return env.agent_states[self.index]

    def _state_attr(name):                                       # noqa: N805  (the reference's PropertyAlias('state', name))
        return property(lambda self: getattr(self.state, name), lambda self, value: setattr(self.state, name, value),
                        doc=f"Alias for `state.{name}` (agent.py:100-109).")

    color = _state_attr("color")
    pos = _state_attr("pos")
    terminated = _state_attr("terminated")
    carrying = _state_attr("carrying")
    del _state_attr

    @property
    def dir(self):
        d = self.state.dir
        return Direction(d) if 0 <= d <= 3 else d

    @dir.setter
    def dir(self, value):
        self.state.dir = value

    @property
    def front_pos(self) -> tuple[int, int]:
        s = self.state
        dx, dy = DIR_TO_VEC[int(s[2])]
        return (int(s[3] + dx), int(s[4] + dy))

    def reset(self, mission: Mission | str = "maximize reward"):
        """agent.py:120-133: a new mission, the state back to "not placed"."""
        self.mission = mission
        st = self.state
        st.pos = (-1, -1)
        st.dir = -1
        st.terminated = False
        st.carrying = None

    def encode(self) -> tuple[int, int, int]:
        s = self.state
        return (int(Type.agent), int(s[1]), int(s[2]))


AgentView = Agent          # (round-1 name)


class GridView:
    """`env.grid` outside `_gen_grid`: the device-resident grid behind the reference's `Grid` accessors -- `state` (W,H,3) int64
    indexed [x, y] (multigrid/core/grid.py:54), `get` (102-117: a `WorldObj` or None), `set` / `update` (78-100, 119-131), which
    write the cell back to the device.  `get` hands back the object the layout placed there while the cell still holds it."""

    def __init__(self, env: "MultiGridEnv"):
        self._env = env

    @property
    def width(self) -> int:
        return self._env.width

    @property
    def height(self) -> int:
        return self._env.height

    def _cells(self) -> np.ndarray:
        """(W,H,3) with a box's content in the upper bits of its state value (include/mgx.h "BOX CONTENTS")"""
        return layouts.grid_from_product(self._env._benv.grid[0].cpu().numpy())

    @property
    def state(self) -> np.ndarray:
        s = self._cells()
        s[..., 2] &= 3                                            # (Grid.state shows no content)
        return s

    @property
    def grid(self) -> list:
        return [self.get(i, j) for i in range(self.width) for j in range(self.height)]

    def get(self, x: int, y: int):
        """The object at (x, y), or None for an empty cell (grid.py:102-117)."""
        if not (0 <= x < self.width and 0 <= y < self.height):
            raise IndexError(f"({x}, {y}) is outside the {self.width}x{self.height} grid")
        return self._env._object_for([int(v) for v in self._cells()[x, y]], pos=(int(x), int(y)))

    def set(self, x: int, y: int, obj):
        """grid.py:78-100: put `obj` (or None) at (x, y) -- on the device."""
        if obj is not None and not isinstance(obj, world.WorldObj):
            raise TypeError(f"cannot set grid value to {type(obj)}")
        cell = list(EMPTY_CELL) if obj is None else list(obj.encode())
        if obj is not None:
            cell[2] |= world.content_code(getattr(obj, "contains", None)) << 2
        self._env._write_cell(x, y, cell)
        self._env._adopt(obj, (int(x), int(y)))

    def update(self, x: int, y: int):
        """grid.py:119-131: write the object known at (x, y) back into the state (after `door.is_open = ...`)."""
        obj = self._env._objects_at.get((int(x), int(y)))
        if obj is not None:
            self.set(x, y, obj)

    def encode(self, vis_mask=None) -> np.ndarray:
        out = self.state
        if vis_mask is not None:
            out[~np.asarray(vis_mask, dtype=bool)] = 0
        return out

    decode = staticmethod(world.Grid.decode)


class MultiGridEnv:
    """Base class.  A subclass supplies the episode start either the reference's way -- `_gen_grid(width, height)`
    (multigrid/base.py:229-247; see the module docstring) -- or as `_gen_layout` (ready-made product tensors: what the
    built-in env classes do), and may override `step` the reference's way (module docstring)."""

    metadata = {"render_modes": [], "render_fps": 20}
    render_mode = None                   # (rendering is out of scope: the constructor refuses any other value)
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
        self._gen_carry_content: dict = {}                # ... and the contents of boxes it hands to agents
        self._objects_at: dict = {}                       # (x, y) -> the WorldObj known to lie there (identity of placed objects)
        self._objects_loose: list = []                    # objects known to the env that are not on the grid (carried)
        self._carried: dict = {}                          # agent index -> THE object it carries, when its identity is known
        self._carry_seen: np.ndarray | None = None        # (A,3) carried cells (type, color, state) as of the last look
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
        self._launched = None                             # what handle_actions' launch rendered, for the `step` that called it

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
    def agent_states(self) -> AgentStateRow:
        """(A,9) int64 snapshot in the reference's AgentState column order (agent.py:222-232) whose setters write through
        (`self.agent_states.terminated = True`, base.py:494).  While `_gen_grid` runs: the episode's initial rows, live -- what
        `RoomGrid._gen_grid` sets (`self.agent_states.pos = ...`, roomgrid.py:232-236) and `reject_next_to` reads (roomgrid.py:45-50)."""
        if self._gen_agents is not None:
            rows = self._gen_agents.view(AgentStateRow)
            rows._env, rows._index, rows._contents = self, None, None
            return rows
        raw = layouts.unpack_agents(self._benv.agents[0].cpu().numpy())
        content = raw[:, 8] >> 2                                 # carried boxes' contents (include/mgx.h): not part of the row
        raw[:, 8] &= 3
        rows = raw.view(AgentStateRow)
        rows._env, rows._index, rows._contents = self, None, content
        return rows

    # -- what the views above write back (one env: small device writes; the dict API is not the throughput path)
    def _write_agents(self, index, cols, value):
        """Columns `cols` of the reference's (A,9) layout for agent `index` (None: every agent) := value, on the device.  Column c
        of that layout (c >= 1) is byte c - 1 of the packed row (include/mgx.h); the tensors are edited in place."""
        ag = self._benv.agents
        which = list(range(9))[cols] if isinstance(cols, slice) else [int(cols)]
        n = self.num_agents if index is None else 1
        v = np.broadcast_to(np.asarray(value, dtype=np.int64), (n, len(which)) if np.ndim(value) else (n, len(which)))
        rows = slice(None) if index is None else slice(index, index + 1)
        for k, c in enumerate(which):
            if c >= 1:
                ag[0, rows, c - 1] = torch.from_numpy((v[:, k] & 0xff).astype(np.uint8)).to(ag.device)

    def _write_cell(self, x: int, y: int, cell):
        """(type, color, state | content << 2) at (x, y), on the device, in the device's cell format."""
        g = np.asarray(cell, dtype=np.uint8).reshape(1, 1, 3)
        packed = layouts.pack_cells_for(self.spec, g)
        self._benv.cells[0, int(y), int(x)] = torch.as_tensor(packed.reshape(()).item(), dtype=self._benv.cells.dtype)

    # -- identity of the objects the layout placed (module docstring)
    def _adopt(self, obj, pos):
        """`obj` now lies at `pos` (None: off the grid, e.g. carried)."""
        for p, o in list(self._objects_at.items()):
            if o is obj or p == pos:
                del self._objects_at[p]
        if obj in self._objects_loose:
            self._objects_loose.remove(obj)
        if obj is None:
            return
        if pos is None:
            self._objects_loose.append(obj)
        else:
            self._objects_at[pos] = obj
            obj.cur_pos = pos

    def _bind_carried(self, index, obj):
        """Agent `index` now carries `obj` (None: nothing): its slot of `_carried`, and what the next look compares against."""
        if index is None:
            return
        if obj is None:
            self._carried.pop(index, None)
        else:
            self._carried[index] = obj
        if self._carry_seen is not None:
            self._carry_seen[index] = (1, 0, 0) if obj is None else [int(v) for v in obj.encode()]

    def _object_for(self, cell, pos, agent=None):
        """The WorldObj for a device cell (type, color, state | content << 2): the known object it identifies, refreshed from the
        cell, or a new one.  `pos`: where the cell lies (None: carried, by agent `agent`).  An object on the grid is identified by
        where it was last seen; a carried one by its carrier -- `_refresh_objects` hands the object that left a cell to the agent
        that stood in front of it and whose hands filled in that step, so two agents carrying look-alikes each hold their own
        (`agent.state.carrying == self.obj`, blockedunlockpickup.py:172, is an identity test).  Only an object whose move was not
        witnessed (state loaded from elsewhere) falls back to being the only displaced object of its type and colour."""
        t, c, sb = int(cell[0]), int(cell[1]), int(cell[2])
        if t == Type.empty:
            if pos is not None and pos in self._objects_at:
                self._objects_loose.append(self._objects_at.pop(pos))     # (it was picked up / replaced)
            return None
        same = lambda o: (int(o.type), int(o.color)) == (t, c)            # noqa: E731
        if pos is None and agent is not None:
            if self._objects_at or self._objects_loose or self._carried:
                self._refresh_objects(scan=False)                         # (binds what was picked up since the last look)
            held = self._carried.get(agent)
            if held is not None and same(held):
                held._v[2] = sb & 3
                if t == Type.box and (sb >> 2) != world.content_code(held.contains):
                    held.contains = world.content_from_code(sb >> 2)
                return held
        known = self._objects_at.get(pos) if pos is not None else None
        if known is None or not same(known):
            displaced = [o for o in self._objects_loose if same(o)]
            if not displaced and any(same(o) for o in self._objects_at.values()):
                self._refresh_objects(scan=False)                         # (not looked at since the step that moved it)
                displaced = [o for o in self._objects_loose if same(o)]
            known = displaced[0] if len(displaced) == 1 else world.WorldObj.from_array(cell)
        known._v[2] = sb & 3
        if t == Type.box and (sb >> 2) != world.content_code(known.contains):
            known.contains = world.content_from_code(sb >> 2)
        self._adopt(known, pos)
        if pos is None and agent is not None:
            self._carried[agent] = known
        return known

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
        self._gen_carry_content = {}
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
            # base.py:286-289: no agent on top of something it could not have walked onto
            for row in ag9:
                start_cell = host.get(int(row[3]), int(row[4]))
                assert start_cell is None or start_cell.can_overlap()
            grid = layouts.grid_to_product(host.state_with_contents())
            layouts.check_walled(grid & np.array([255, 255, 3], dtype=np.uint8))   # the kernels' precondition (include/mgx.h)
            rows = layouts.pack_agents(ag9)
            for i, code in self._gen_carry_content.items():
                rows[i, 7] |= code << 2
            # the objects the layout placed keep their identity for the episode (module docstring)
            self._objects_at = {p: o for p, o in host.world_objects.items() if o is not None
                                and tuple(int(v) for v in host.state[p][:2]) == (int(o.type), int(o.color))}
            self._objects_loose = []
            return grid, rows, None
        finally:
            self._gen_agents = None
            self.grid = live_grid                                             # `env.grid` shows the device-resident state again

    # ------------------------------------------------------------------------------------ the reference's extension point
    def _gen_grid(self, width: int, height: int):
        """multigrid/base.py:229-247: generate the grid for a new episode -- set `self.grid` and populate it with `WorldObj`s,
        set the position and direction of every agent.  Override it in a subclass exactly as with the reference."""
        raise NotImplementedError

    def put_obj(self, obj, i: int, j: int):
        """`obj` at the fixed cell (i, j); the object remembers where it started (base.py:659-665)."""
        where = (i, j)
        self.grid.set(*where, obj)
        obj.init_pos = obj.cur_pos = where

    def _free_for_placement(self, cell, reject_fn) -> bool:
        """May `place_obj` use `cell`?  Nothing lies there, no agent stands there, the caller's rule does not object."""
        if self.grid.get(*cell) is not None:
            return False
        rows = self._gen_agents
        if bool(((rows[:, 3] == cell[0]) & (rows[:, 4] == cell[1])).any()):
            return False
        return not (reject_fn is not None and reject_fn(self, cell))

    def place_obj(self, obj, top=None, size=None, reject_fn=None, max_tries=float("inf")):
        """Rejection sampling of a free cell inside the rectangle (`top`, `size`), default the whole grid (base.py:604-657).  Two
        draws per attempt -- x then y, each `_rand_int(lo, min(lo + extent, grid side))` from the construction-time generator
         -- so the sequence of draws, and with it the layout, is the reference's."""
        if self._gen_agents is None:
            raise RuntimeError("place_obj edits the episode being generated: it is available inside _gen_grid only")
        grid = self.grid
        x0, y0 = (0, 0) if top is None else (max(int(top[0]), 0), max(int(top[1]), 0))
        w, h = (grid.width, grid.height) if size is None else size
        x1, y1 = min(x0 + w, grid.width), min(y0 + h, grid.height)
        attempts = 0
        while True:
            if attempts > max_tries:                                         # (the reference's exception type, base.py:634-635)
                raise RecursionError(f"place_obj: no free cell found in {attempts} attempts")
            attempts += 1
            cell = (self._rand_int(x0, x1), self._rand_int(y0, y1))
            if self._free_for_placement(cell, reject_fn):
                break
        grid.set(*cell, obj)
        if obj is not None:
            obj.init_pos = obj.cur_pos = cell
        return cell

    def place_agent(self, agent, top=None, size=None, rand_dir=True, max_tries=float("inf")):
        """The agent on a free cell of the rectangle, facing a random direction unless told otherwise (base.py:667-686).  The agent
        is taken off the grid first so that its old cell is free for the draw."""
        agent.state.pos = (-1, -1)
        cell = self.place_obj(None, top, size, max_tries=max_tries)
        agent.state.pos = cell
        if rand_dir:
            agent.state.dir = self._rand_int(0, 4)
        return cell

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
        """`num_elems` distinct picks, one `_rand_int` draw each over what is left (random.py:60-74).  A pick takes the FIRST
        remaining element equal to the drawn one out of the pool, which is what keeps later draws aligned with the reference when
        the pool holds duplicates."""
        pool = list(iterable)
        assert num_elems <= len(pool)
        picked = []
        for _ in range(num_elems):
            choice = pool[self._rand_int(0, len(pool))]
            pool.remove(choice)
            picked.append(choice)
        return picked

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
        self._objects_at, self._objects_loose, self._carried, self._carry_seen = {}, [], {}, None
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
        """multigrid/base.py:303-346.  Returns (observations, rewards, terminations, truncations, infos).  A subclass may extend
        it the reference's way (module docstring): the dicts returned here are plain dicts to update."""
        A = self.num_agents
        # base.py:333 `self.step_count += 1` -- the fused kernel does it; only a subclass that REPLACES handle_actions (and may read
        # `self.step_count` / `_reward()` inside it) needs the count advanced before the call
        custom_actions = type(self).handle_actions is not MultiGridEnv.handle_actions
        if custom_actions:
            self._benv.step_count += 1
        self._launched = None
        self._count_advanced = custom_actions
        rewards = self.handle_actions(actions)                           # base.py:334
        if self._launched is None:                # a subclass replaced handle_actions wholesale: render what it left (base.py:337)
            if not custom_actions:
                self._benv.step_count += 1
            obs, dirs = self._benv.gen_obs()
            host = self._benv.outputs_to_host()
            obs, dirs = host["obs"][0].copy(), host["dir"][0].copy()
            term = self._benv.agents[0, :, 4].cpu().numpy().reshape(-1)
            truncated = self.step_count >= self.max_steps
        else:                                     # the fused kernel rendered the post-action state in the same launch; its outputs
            host = self._launched                 # came over in handle_actions' ONE device-to-host copy
            obs, dirs, term = host["obs"][0].copy(), host["dir"][0].copy(), host["terminated"][0].copy()
            truncated = bool(host["truncated"][0])                       # base.py:339
        terminations = {i: bool(term[i]) for i in range(A)}              # base.py:338
        truncations = {i: truncated for i in range(A)}
        if self._objects_at or self._objects_loose or self._carried:
            self._refresh_objects()
        return self._obs_dict(obs, dirs), rewards, terminations, truncations, defaultdict(dict)

    def handle_actions(self, actions: dict[int, int]) -> dict:
        """multigrid/base.py:378-476: the agents act in random order; returns the rewards dict.  Here it IS the fused kernel's
        launch (multigrid_amd/csrc/mgx_fused.h), which applies the actions against the step count `step` has already advanced
        (base.py:333; `_reward` reads it, base.py:598-602) and renders the observations `step` returns in the same launch."""
        A = self.num_agents
        act = np.full((1, A), NO_ACTION, dtype=np.int8)
        keys = []
        if not hasattr(actions, "items"):
            # a sequence instead of a dict: the reference's loop reads `if i not in actions: continue; action = actions[i]`
            # (base.py:402-406) -- for a list that is a test on the VALUES: agent i acts iff the number i is among them, and then takes
            # actions[i].  `step([2, 2])` therefore moves nobody.
            actions = {i: actions[i] for i in range(A) if i in actions}
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
        if getattr(self, "_count_advanced", False):
            benv.step_count -= 1                  # (the kernel's own `step_count += 1` is the one `step` has already done)
            self._count_advanced = False
        benv.step(torch.from_numpy(act).to(benv.device), hook_order=hook_order)
        host = benv.outputs_to_host()             # ONE copy: reward, obs, dir, terminated, truncated, the error words
        if int(host["err"][0]):
            benv._reset_err()
            bad = [int(a) for a in actions.values() if not 0 <= int(a) <= int(Action.done)]
            raise ValueError(f"Unknown action: {bad[0] if bad else '?'}")          # base.py:473-474
        self._launched = host
        rew = host["reward"][0]
        return {i: (float(rew[i]) if rew[i] != 0 else 0) for i in range(A)}       # int 0 unless rewarded (base.py:393)

    def _refresh_objects(self, scan: bool = True):
        """After a step: the objects the layout placed follow what the kernel did to their cells (a door's state, a box's content,
        an object picked up or dropped), so that a `step` override reads them as the reference's hook reads its objects."""
        cells = self.grid._cells()
        # who picked up / put down what since the last look: an agent whose hands filled took the object of the cell in front of it
        # (pickup leaves the agent where it stood, base.py:436-447), one whose hands emptied left its object there (base.py:449-462)
        rows = layouts.unpack_agents(self._benv.agents[0].cpu().numpy())
        now = rows[:, 6:9].copy()
        now[:, 2] &= 3
        seen = self._carry_seen if self._carry_seen is not None else np.tile(np.array([1, 0, 0]), (self.num_agents, 1))
        for i in range(self.num_agents):
            had, has = int(seen[i, 0]) != int(Type.empty), int(now[i, 0]) != int(Type.empty)
            if had == has and (not has or tuple(seen[i, :2]) == tuple(now[i, :2])):
                continue
            dx, dy = DIR_TO_VEC[int(rows[i, 2]) & 3]
            front = (int(rows[i, 3] + dx), int(rows[i, 4] + dy))
            if had:                                                      # put down (or swapped by a user hook): it lies in front
                obj = self._carried.pop(i, None)
                if obj is not None and not has and tuple(int(v) for v in cells[front][:2]) == (int(obj.type), int(obj.color)):
                    self._adopt(obj, front)
            if has:                                                      # picked up: the object that lay in front, if known
                obj = self._objects_at.get(front)
                if obj is not None and (int(obj.type), int(obj.color)) == (int(now[i, 0]), int(now[i, 1])) \
                        and int(cells[front][0]) != int(obj.type):
                    del self._objects_at[front]
                    obj.cur_pos = None
                    self._carried[i] = obj
        self._carry_seen = now
        for pos, obj in list(self._objects_at.items()):
            cell = [int(v) for v in cells[pos]]
            if (cell[0], cell[1]) == (int(obj.type), int(obj.color)):
                obj._v[2] = cell[2] & 3
            else:                                                    # gone from there: picked up, or the box was opened
                del self._objects_at[pos]
                self._objects_loose.append(obj)
                obj.cur_pos = None
        if scan and self._objects_loose:                                 # dropped somewhere else?
            for x, y in np.argwhere(cells[..., 0] > int(Type.floor)):
                p = (int(x), int(y))
                if p not in self._objects_at:
                    self._object_for([int(v) for v in cells[p]], pos=p)

    # ------------------------------------------------------------------------------------ episode-ending callbacks
    def _reward(self) -> float:
        """base.py:598-602"""
        return 1 - 0.9 * (self.step_count / self.max_steps)

    def on_success(self, agent: Agent, rewards: dict, terminations: dict):
        """base.py:478-507: callback for when `agent` completes its mission -- it (or, in mode 'any', every agent) is terminated, on
        the device too; it (or, with `joint_reward`, every agent) is paid `_reward()`."""
        if self.success_termination_mode == "any":
            self.agent_states.terminated = True
            for i in range(self.num_agents):
                terminations[i] = True
        else:
            agent.state.terminated = True
            terminations[agent.index] = True
        if self.joint_reward:
            for i in range(self.num_agents):
                rewards[i] = self._reward()
        else:
            rewards[agent.index] = self._reward()

    def on_failure(self, agent: Agent, rewards: dict, terminations: dict):
        """base.py:509-532: callback for when `agent` fails its mission prematurely."""
        if self.failure_termination_mode == "any":
            self.agent_states.terminated = True
            for i in range(self.num_agents):
                terminations[i] = True
        else:
            agent.state.terminated = True
            terminations[agent.index] = True

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
