"""User-defined environments written against the reference's extension point, `MultiGridEnv._gen_grid(width, height)`
(multigrid/base.py:229-247).  The SAME class bodies run on both implementations:

    define(multigrid_namespace())        -> classes over ini/multigrid         (oracle/gen_golden.py, build container only)
    define(multigrid_amd_namespace())    -> classes over multigrid_amd         (tests/, CPU oracle backend and GPU)

and tests/golden/custom_*.npz holds what the reference produced (reset sequences: grid, agent states, np_random, first
observations).  The env definitions are this repo's own; nothing here comes from the reference's sources -- they only USE its
public API (`Grid.wall_rect / horz_wall / vert_wall / set / get`, `put_obj`, `place_obj`, `place_agent`, `_rand_*`, the `WorldObj`
classes, `agent.state.pos / dir`).
"""
from types import SimpleNamespace


def multigrid_amd_namespace():
    import multigrid_amd as m
    from multigrid_amd import core
    from multigrid_amd.core.roomgrid import RoomGrid
    return SimpleNamespace(MultiGridEnv=m.MultiGridEnv, Grid=core.Grid, Goal=core.Goal, Wall=core.Wall, Door=core.Door, Key=core.Key,
                           Ball=core.Ball, Box=core.Box, Floor=core.Floor, Lava=core.Lava, Color=core.Color, Direction=core.Direction,
                           Action=core.Action, RoomGrid=RoomGrid, Type=core.Type)


def multigrid_namespace():
    from multigrid.base import MultiGridEnv
    from multigrid import core
    from multigrid.core.roomgrid import RoomGrid
    return SimpleNamespace(MultiGridEnv=MultiGridEnv, Grid=core.Grid, Goal=core.Goal, Wall=core.Wall, Door=core.Door, Key=core.Key,
                           Ball=core.Ball, Box=core.Box, Floor=core.Floor, Lava=core.Lava, Color=core.Color, Direction=core.Direction,
                           Action=core.Action, RoomGrid=RoomGrid, Type=core.Type)


def define(ns):
    """The custom env classes over the implementation `ns` names."""

    class TwoRoomsEnv(ns.MultiGridEnv):
        """Two rooms split by a wall with a locked yellow door; the key lies somewhere in the left room, the goal in the right
        one behind a strip of lava with a gap; a ball and an (empty) box are scattered with a rejection rule; agent 0 starts at a
        fixed cell, the others are placed at random in the left room."""

        def __init__(self, size=11, **kwargs):
            super().__init__(mission_space="unlock the door and reach the goal", grid_size=size, max_steps=6 * size * size,
                             **kwargs)

        def _gen_grid(self, width, height):
            self.grid = ns.Grid(width, height)
            self.grid.wall_rect(0, 0, width, height)
            mid = width // 2
            self.grid.vert_wall(mid, 0)                                  # to the bottom edge
            door_y = self._rand_int(1, height - 1)
            self.door = ns.Door(ns.Color.yellow, is_locked=True)
            self.put_obj(self.door, mid, door_y)
            # a lava strip in the right room, two cells before the goal column, with a one-cell gap
            gap = self._rand_int(1, height - 1)
            self.grid.vert_wall(width - 3, 1, height - 2, obj_type=ns.Lava)
            self.grid.set(width - 3, gap, None)
            self.put_obj(ns.Goal(), width - 2, height - 2)
            self.put_obj(ns.Floor(ns.Color.purple), mid + 1, 1)
            # agent 0 at a fixed start, the others anywhere in the left room
            for agent in self.agents:
                if agent.index == 0:
                    agent.state.pos = (1, 1)
                    agent.state.dir = ns.Direction.down
                else:
                    self.place_agent(agent, top=(1, 1), size=(mid - 1, height - 2))
            self.key_pos = self.place_obj(ns.Key(ns.Color.yellow), top=(1, 1), size=(mid - 1, height - 2))
            # a ball of a random colour, never on an even column; a box not next to the door row
            self.place_obj(ns.Ball(self._rand_color()), reject_fn=lambda env, pos: pos[0] % 2 == 0)
            self.place_obj(ns.Box(ns.Color.green), top=(mid + 1, 1), size=(2, height - 2),
                           reject_fn=lambda env, pos: abs(pos[1] - door_y) <= 1, max_tries=1000)

    class ScatterEnv(ns.MultiGridEnv):
        """A non-square arena with an inner horizontal wall segment, closed and open doors, and objects drawn with the whole
        `_rand_*` family; every agent placed at random with a random direction."""

        def __init__(self, width=13, height=9, **kwargs):
            super().__init__(mission_space="wander", width=width, height=height, max_steps=200, **kwargs)

        def _gen_grid(self, width, height):
            self.grid = ns.Grid(width, height)
            self.grid.wall_rect(0, 0, width, height)
            y = height // 2
            self.grid.horz_wall(2, y, width - 4)
            xs = self._rand_subset(range(3, width - 3), 2)
            self.put_obj(ns.Door(self._rand_color(), is_open=self._rand_bool()), xs[0], y)
            self.put_obj(ns.Door(ns.Color.red), xs[1], y)
            for color in self._rand_perm([ns.Color.blue, ns.Color.green, ns.Color.grey]):
                self.place_obj(ns.Key(color), top=(1, 1), size=(width - 2, y - 1))
            x, yy = self._rand_pos(1, width - 1, y + 1, height - 1)
            if self.grid.get(x, yy) is None:
                self.put_obj(ns.Lava(), x, yy)
            self.place_obj(ns.Ball(self._rand_elem([ns.Color.purple, ns.Color.yellow])), top=(1, y + 1), size=(width - 2, height - y - 2))
            self.put_obj(ns.Goal(), width - 2, 1)
            for agent in self.agents:
                self.place_agent(agent)

    class BoxTreasureEnv(ns.MultiGridEnv):
        """Boxes that HOLD things (Box(color, contains=...), multigrid/core/world_object.py:574-605): the key of the locked door
        between the two rooms is in a box, other boxes hold a ball, a goal, a closed door, nothing.  Toggling a box replaces it by
        its content; a box can be carried away first and opened elsewhere.  Agents are placed through the `Agent` aliases
        (`agent.pos = ...`, `agent.dir = ...`, agent.py:100-109)."""

        def __init__(self, size=9, **kwargs):
            super().__init__(mission_space="open the boxes", grid_size=size, max_steps=5 * size * size, **kwargs)

        def _gen_grid(self, width, height):
            self.grid = ns.Grid(width, height)
            self.grid.wall_rect(0, 0, width, height)
            mid = width // 2
            self.grid.vert_wall(mid, 0)
            self.door = ns.Door(ns.Color.purple, is_locked=True)
            self.put_obj(self.door, mid, height // 2)
            self.put_obj(ns.Goal(), width - 2, 1)
            self.key_box = ns.Box(ns.Color.yellow, contains=ns.Key(ns.Color.purple))
            self.put_obj(self.key_box, 1, height - 2)
            self.put_obj(ns.Box(ns.Color.red, contains=ns.Ball(ns.Color.blue)), 2, 1)
            self.put_obj(ns.Box(ns.Color.green), 3, 3)                                   # holds nothing
            self.place_obj(ns.Box(self._rand_color(), contains=ns.Goal(ns.Color.grey)), top=(1, 1), size=(mid - 1, height - 2))
            self.place_obj(ns.Box(ns.Color.blue, contains=ns.Door(ns.Color.red)), top=(mid + 1, 1), size=(mid - 1, height - 2))
            self.place_obj(ns.Box(ns.Color.grey, contains=ns.Lava()), top=(mid + 1, 1), size=(mid - 1, height - 2))
            for agent in self.agents:
                if agent.index == 0:
                    agent.pos = (1, height - 3)                                          # right above the key box, facing it
                    agent.dir = ns.Direction.down
                else:
                    self.place_agent(agent, top=(1, 1), size=(mid - 1, height - 2))

    class FetchTrapEnv(ns.MultiGridEnv):
        """An env that ends its episodes the way the reference's own envs do (envs/blockedunlockpickup.py:166-175,
        envs/redbluedoors.py:170-187): a `step` override on top of the base step -- success for whoever carries THE purple ball
        (object identity, not the green decoy), failure for whoever toggles the red trap door open."""

        def __init__(self, size=8, **kwargs):
            super().__init__(mission_space="fetch the purple ball, leave the red door shut", grid_size=size,
                             max_steps=3 * size * size, **kwargs)

        def _gen_grid(self, width, height):
            self.grid = ns.Grid(width, height)
            self.grid.wall_rect(0, 0, width, height)
            self.grid.horz_wall(0, height - 3)
            self.trap = ns.Door(ns.Color.red)
            self.put_obj(self.trap, self._rand_int(1, width - 1), height - 3)
            self.ball = ns.Ball(ns.Color.purple)
            self.place_obj(self.ball, top=(1, 1), size=(width - 2, height - 4))
            self.decoy = ns.Ball(ns.Color.green)
            self.place_obj(self.decoy, top=(1, 1), size=(width - 2, height - 4))
            for agent in self.agents:
                self.place_agent(agent, top=(1, 1), size=(width - 2, height - 4))

        def step(self, actions):
            obs, reward, terminated, truncated, info = super().step(actions)
            for agent in self.agents:
                if agent.state.carrying == self.ball:
                    self.on_success(agent, reward, terminated)
            for agent_id, action in actions.items():
                if action == ns.Action.toggle:
                    agent = self.agents[agent_id]
                    if self.grid.get(*agent.front_pos) == self.trap and self.trap.is_open:
                        self.on_failure(agent, reward, terminated)
            return obs, reward, terminated, truncated, info

    class VaultRoomsEnv(ns.RoomGrid):
        """A `RoomGrid` subclass (multigrid/core/roomgrid.py:139-495), written the way the reference's own room envs are: 2 x 3 rooms;
        the vault -- a box of a random colour -- lies in the far corner room behind a LOCKED door whose key lies in the start room; one
        inner wall is taken out, `connect_all` adds what doors are still needed, `add_distractors` adds what it adds (one object:
        the reference's bookkeeping fails after the first, roomgrid.py:493), every agent starts in room (0, 0).  The `step` override
        ends the episode for whoever carries THE vault (its type compared the reference's way: `obj.type == 'box'`)."""

        def __init__(self, room_size=5, **kwargs):
            super().__init__(room_size=room_size, num_rows=2, num_cols=3, mission_space="carry the vault box",
                             max_steps=8 * room_size ** 2, **kwargs)

        def _gen_grid(self, width, height):
            super()._gen_grid(width, height)
            self.vault, _ = self.add_object(2, 1, kind=ns.Type.box)
            self.vault_door, _ = self.add_door(2, 1, ns.Direction.left, locked=True)
            self.add_object(0, 0, ns.Type.key, self.vault_door.color)
            # the room in front of the vault door counts as locked too (Room.locked): connect_all will not touch it, so it gets its
            # way out by hand; the start room gets a grey door in the middle of its right wall
            self.add_door(1, 1, ns.Direction.up, color=ns.Color.blue, locked=False)
            self.add_door(0, 0, ns.Direction.right, color=ns.Color.grey, locked=False, rand_pos=False)
            self.extra_doors = self.connect_all(door_colors=[ns.Color.purple, ns.Color.yellow])
            try:
                self.add_distractors(1, 0, num_distractors=3)
            except AttributeError:
                pass
            self.n_distractors = len(self.get_room(1, 0).objs)
            self.add_object(0, 1)                                       # a random key / ball / box of a random colour
            for agent in self.agents:
                self.place_agent(agent, 0, 0)
            assert self.room_from_pos(*self.vault.init_pos) is self.get_room(2, 1) and self.get_room(2, 1).locked
            assert self.get_room(0, 0).pos_inside(1, 1) and not self.get_room(0, 0).pos_inside(width - 2, 1)
            # (last: `Room.locked` cannot look at a room with a removed wall, roomgrid.py:85)
            if self.get_room(0, 0).doors[ns.Direction.down] is None:
                self.remove_wall(0, 0, ns.Direction.down)

        def step(self, actions):
            obs, reward, terminated, truncated, info = super().step(actions)
            for agent in self.agents:
                held = agent.state.carrying
                if held is not None and held.type == 'box' and held == self.vault:
                    self.on_success(agent, reward, terminated)
            return obs, reward, terminated, truncated, info

    class TwinBallsEnv(ns.MultiGridEnv):
        """Two balls that LOOK alike -- same type, same colour -- of which only one counts: `agent.state.carrying == self.prize` is an
        identity test (multigrid/core/world_object.py:126-127), so when both agents hold a purple ball at the same time exactly one of
        them has succeeded.  Agent 0 starts in front of the twin, agent 1 in front of the prize; success mode 'all', so the other agent
        plays on -- putting its ball down, picking either one up again."""

        def __init__(self, size=7, **kwargs):
            super().__init__(mission_space="fetch the right purple ball", grid_size=size, max_steps=4 * size * size,
                             success_termination_mode="all", **kwargs)

        def _gen_grid(self, width, height):
            self.grid = ns.Grid(width, height)
            self.grid.wall_rect(0, 0, width, height)
            self.twin, self.prize = ns.Ball(ns.Color.purple), ns.Ball(ns.Color.purple)
            self.put_obj(self.twin, 1, 2)
            self.put_obj(self.prize, 3, 2)
            self.place_obj(ns.Ball(ns.Color.purple), top=(1, 3), size=(width - 2, height - 4))      # a third look-alike, somewhere
            for agent in self.agents:
                agent.state.pos = (1 + 2 * agent.index, 1)
                agent.state.dir = ns.Direction.down

        def step(self, actions):
            obs, reward, terminated, truncated, info = super().step(actions)
            for agent in self.agents:
                if agent.state.carrying == self.prize:
                    self.on_success(agent, reward, terminated)
            return obs, reward, terminated, truncated, info

    return {"TwoRoomsEnv": TwoRoomsEnv, "ScatterEnv": ScatterEnv, "BoxTreasureEnv": BoxTreasureEnv, "FetchTrapEnv": FetchTrapEnv,
            "VaultRoomsEnv": VaultRoomsEnv, "TwinBallsEnv": TwinBallsEnv}


def intervene(cname, env, t):
    """What the recording script does to a running episode BESIDES stepping it -- the same call is made by the recorder (over the
    reference) and by the replaying tests (over multigrid_amd) right before step `t`.  Returns {agent index: action} to play at this
    step instead of the random ones (the fixtures hold the actions as played), or None.

    VaultRoomsEnv, t = 60: agent 0 is put down next to the vault, facing it, with empty hands (through the state setters:
    `agent.state.pos / dir / carrying = ...`, multigrid/core/agent.py:286-346) and picks it up -- so that the `step` override's
    success path is part of what was recorded (a random walk does not get through the locked door in time)."""
    if cname == "VaultRoomsEnv" and t == 60:
        vx, vy = (int(v) for v in env.vault.init_pos)
        box = env.grid.get(vx, vy)
        assert box is not None and box.type == 'box', "the vault has not moved"
        taken = {tuple(int(v) for v in a.state.pos) for a in env.agents if a.index != 0}
        for d, (dx, dy) in enumerate(((-1, 0), (0, -1), (1, 0), (0, 1))):      # stand left of it facing right, above facing down, ...
            cell = (vx + dx, vy + dy)
            if env.grid.get(*cell) is None and cell not in taken:
                agent = env.agents[0]
                agent.state.carrying = None
                agent.state.pos = cell
                agent.state.dir = d
                return {0: 3}                                                   # Action.pickup
        raise AssertionError("no free cell next to the vault")
    if cname == "TwinBallsEnv":
        # step 0: both agents pick up the ball in front of them -- agent 0 the twin, agent 1 the prize; from step 12 on agent 1 (the
        # one that is done) is left out of the dict while agent 0 swaps balls around: the actions as drawn
        if t == 0:
            return {0: 3, 1: 3}
        if t in (3, 4, 5):                                                      # agent 0: put the twin down, turn away and back
            return {0: (4, 0, 1)[t - 3]}
    return None


#: fixture name -> (class name, constructor kwargs)
CASES = {
    "custom_tworooms_a3": ("TwoRoomsEnv", dict(size=11, agents=3)),
    "custom_scatter_a2_v5": ("ScatterEnv", dict(width=13, height=9, agents=2, agent_view_size=5, allow_agent_overlap=False)),
}
#: round 6 -- recorded BEHIND every older fixture (oracle/gen_golden.py gives each env it makes the next construction seed, so the
#: older fixtures keep theirs and regenerate byte for byte): a RoomGrid subclass
CASES_R6 = {
    "custom_vaultrooms_a2": ("VaultRoomsEnv", dict(room_size=5, agents=2)),
    "custom_vaultrooms_a3_rs6": ("VaultRoomsEnv", dict(room_size=6, agents=3, agent_view_size=5)),
}

#: step-sequence fixtures (oracle/gen_golden.py: record_custom_steps; tests/test_custom_envs.py: _replay_steps):
#: fixture name -> (class name, constructor kwargs, steps per episode)
STEP_CASES = {
    "customsteps_boxtreasure_a3": ("BoxTreasureEnv", dict(size=9, agents=3), 120),
    "customsteps_fetchtrap_a2": ("FetchTrapEnv", dict(size=8, agents=2), 90),
    "customsteps_fetchtrap_a3_all": ("FetchTrapEnv", dict(size=8, agents=3, success_termination_mode="all",
                                                          failure_termination_mode="any", joint_reward=True), 90),
}
STEP_CASES_R6 = {
    "customsteps_vaultrooms_a2": ("VaultRoomsEnv", dict(room_size=5, agents=2, joint_reward=True), 150),
    "customsteps_twinballs_a2": ("TwinBallsEnv", dict(size=7, agents=2), 70),
    "customsteps_twinballs_a3_joint": ("TwinBallsEnv", dict(size=9, agents=3, joint_reward=True), 90),
}
ALL_CASES = {**CASES, **CASES_R6}
ALL_STEP_CASES = {**STEP_CASES, **STEP_CASES_R6}
