"""The reference's extension point: a user-defined env that overrides `_gen_grid(width, height)` (multigrid/base.py:229-247) and
builds its episode with `Grid.wall_rect / horz_wall / vert_wall / set / get` (core/grid.py:78-195), `put_obj` / `place_obj` /
`place_agent` (base.py:604-697), the `_rand_*` family (utils/random.py:9-103) and the WorldObj classes (core/world_object.py:
279-616) runs UNCHANGED on multigrid_amd.  tests/custom_envs.py holds the class bodies (this repo's own); oracle/gen_golden.py ran
them over the real reference and recorded tests/golden/custom_*.npz; here the same bodies run over multigrid_amd -- on the CPU
oracle backend (host logic) and on the GPU (HIP kernels) -- and must reproduce every reset: grid, agent states, np_random,
observations, and the five steps after it."""
import os

import numpy as np
import pytest

import multigrid_amd as mg
from multigrid_amd import core
from tests import custom_envs, util


def _replay(path, **env_kw):
    z = np.load(path)
    cname, kw = custom_envs.ALL_CASES[os.path.basename(path)[:-4]]
    cls = custom_envs.define(custom_envs.multigrid_amd_namespace())[cname]
    env = cls(layout_seed=int(z["construct_seed"]), **kw, **env_kw)
    A = env.num_agents
    for k, sd in enumerate(z["reset_seeds"]):
        obs, infos = env.reset(seed=None if sd < 0 else int(sd))
        ctx = f"reset {k}"
        np.testing.assert_array_equal(env.grid.state, z["grid0"][k].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(env.agent_states, z["agents0"][k].astype(np.int64), err_msg=ctx)
        got = env._benv.rng[0].cpu().numpy().view(np.uint64)
        np.testing.assert_array_equal(got, util.rng_words_lohi(z["rng0"][k]), err_msg=ctx)
        for i in range(A):
            np.testing.assert_array_equal(obs[i]["image"], z["obs0"][k][i], err_msg=ctx)
        assert env.step_count == 0
        for t in range(5):
            o, r, tm, tr, _ = env.step({i: int(z["actions"][k][t, i]) for i in range(A)})
            for i in range(A):
                np.testing.assert_array_equal(o[i]["image"], z["obs_steps"][k][t][i], err_msg=f"{ctx} step {t}")
                assert float(r[i]) == float(z["reward_steps"][k][t][i]) and bool(tm[i]) == bool(z["terminated_steps"][k][t][i])
    return env


@pytest.mark.parametrize("path", util.CUSTOM_GOLDEN, ids=util.CUSTOM_IDS)
def test_user_defined_gen_grid_matches_reference_on_cpu(path):
    env = _replay(path, device="cpu", _backend=lambda spec: util.OracleBackend(spec))
    assert isinstance(env.grid, mg.env.GridView)                   # `env.grid` shows the live (device-resident) state again


@pytest.mark.gpu
@pytest.mark.parametrize("path", util.CUSTOM_GOLDEN, ids=util.CUSTOM_IDS)
def test_user_defined_gen_grid_matches_reference_on_gpu(path):
    _replay(path, device="cuda")


def _replay_steps(path, **env_kw):
    """Whole episodes of a user-defined env (tests/custom_envs.py: STEP_CASES) against what the REAL reference did with the same
    class body: every step's observations, rewards, terminations, truncations, and the post-step grid and agent rows -- box
    contents included (the fixture folds them into the state values, include/mgx.h "BOX CONTENTS")."""
    from multigrid_amd import layouts
    z = np.load(path)
    cname, kw, T = custom_envs.ALL_STEP_CASES[os.path.basename(path)[:-4]]
    cls = custom_envs.define(custom_envs.multigrid_amd_namespace())[cname]
    env = cls(layout_seed=int(z["construct_seed"]), **kw, **env_kw)
    A = env.num_agents
    n_reward = n_term = 0
    for k, sd in enumerate(z["reset_seeds"]):
        obs, _ = env.reset(seed=None if sd < 0 else int(sd))
        ctx = f"episode {k}"
        np.testing.assert_array_equal(env.grid._cells(), z["grid0"][k].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(layouts.unpack_agents(env._benv.agents[0].cpu().numpy()), z["agents0"][k].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(env.grid.state, z["grid0"][k].astype(np.int64) & np.array([255, 255, 3]), err_msg=ctx)
        for i in range(A):
            np.testing.assert_array_equal(obs[i]["image"], z["obs0"][k][i], err_msg=ctx)
        for t in range(T):
            custom_envs.intervene(cname, env, t)                       # (what the recorder did to the episode besides stepping it)
            acts = {i: int(z["actions"][k][t, i]) for i in range(A) if z["actions"][k][t, i] >= 0}
            o, r, tm, tr, _ = env.step(acts)
            c = f"{ctx} step {t}"
            for i in range(A):
                np.testing.assert_array_equal(o[i]["image"], z["obs"][k][t][i], err_msg=c)
                assert float(r[i]) == float(z["reward"][k][t][i]), (c, i, r, z["reward"][k][t])
                assert bool(tm[i]) == bool(z["terminated"][k][t][i]), (c, i, tm, z["terminated"][k][t])
                assert bool(tr[i]) == bool(z["truncated"][k][t]), c
            np.testing.assert_array_equal(env.grid._cells(), z["grid"][k][t].astype(np.int64), err_msg=c)
            np.testing.assert_array_equal(layouts.unpack_agents(env._benv.agents[0].cpu().numpy()), z["agents"][k][t].astype(np.int64),
                                          err_msg=c)
            n_reward += any(float(r[i]) > 0 for i in range(A)); n_term += any(bool(tm[i]) for i in range(A))
    return env, n_reward, n_term


@pytest.mark.parametrize("path", util.CUSTOM_STEPS_GOLDEN, ids=util.CUSTOM_STEPS_IDS)
def test_user_defined_step_hooks_and_box_contents_match_reference_on_cpu(path):
    env, n_reward, n_term = _replay_steps(path, device="cpu", _backend=lambda spec: util.OracleBackend(spec))
    if "fetchtrap" in path:
        assert n_reward > 10 and n_term > 10                       # the `step` override's on_success / on_failure did fire


@pytest.mark.gpu
@pytest.mark.parametrize("path", util.CUSTOM_STEPS_GOLDEN, ids=util.CUSTOM_STEPS_IDS)
def test_user_defined_step_hooks_and_box_contents_match_reference_on_gpu(path):
    _replay_steps(path, device="cuda")


def test_boxes_hold_things_as_in_the_reference():
    """multigrid/core/world_object.py:574-605"""
    key = core.Key("purple")
    box = core.Box("yellow", contains=key)
    assert box.contains is key and box.encode() == (core.Type.box, core.Color.yellow, 0)          # the encoding does not show it
    g = core.Grid(5, 5)
    g.set(2, 2, box)
    assert g.state[2, 2].tolist() == [7, 4, 0] and g.state_with_contents()[2, 2].tolist() == [7, 4, (1 | 3 << 3) << 2]
    back = core.WorldObj.from_array(g.state_with_contents()[2, 2])
    assert isinstance(back, core.Box) and isinstance(back.contains, core.Key) and back.contains.color == core.Color.purple
    assert core.Box(contains=core.Door("red")).contains.state == core.State.closed
    with pytest.raises(NotImplementedError, match="box inside a box"):
        core.Box(contains=core.Box())
    with pytest.raises(NotImplementedError, match="closed and unlocked"):
        core.Box(contains=core.Door("red", is_locked=True))

    class Env:                                                       # Box.toggle / Door.toggle on a host grid (world_object.py:215)
        grid = g
    assert box.toggle(Env, None, (2, 2)) is True and g.get(2, 2) is key


def test_world_objects_have_the_reference_encodings_and_predicates():
    """multigrid/core/world_object.py:279-616 (constructor defaults, encodings) and :197-233 + overrides (predicates)."""
    C, T, S = core.Color, core.Type, core.State
    assert core.Goal().encode() == (T.goal, C.green, 0) and core.Floor().encode() == (T.floor, C.blue, 0)
    assert core.Lava().encode() == (T.lava, C.red, 0) and core.Wall().encode() == (T.wall, C.grey, 0)
    assert core.Key().encode() == (T.key, C.blue, 0) and core.Ball("red").encode() == (T.ball, C.red, 0)
    assert core.Box().encode() == (T.box, C.yellow, 0)
    assert core.Door().encode() == (T.door, C.blue, S.closed)
    assert core.Door("red", is_open=True).encode() == (T.door, C.red, S.open)
    assert core.Door(C.grey, is_locked=True).encode() == (T.door, C.grey, S.locked)
    d = core.Door(is_open=True, is_locked=True)                     # the setters run in this order (world_object.py:398-413)
    assert d.state == S.locked and not d.is_open
    d.is_locked = False
    assert d.state == S.closed
    d.is_open = True
    assert d.can_overlap() and not core.Door().can_overlap()
    assert core.Goal().can_overlap() and core.Floor().can_overlap() and core.Lava().can_overlap() and not core.Wall().can_overlap()
    assert core.Key().can_pickup() and core.Ball().can_pickup() and core.Box().can_pickup() and not core.Goal().can_pickup()
    assert core.Box().can_contain() and not core.Key().can_contain()
    assert isinstance(core.WorldObj(type="goal", color="green"), core.Goal)
    assert core.WorldObj.from_array([1, 0, 0]) is None and isinstance(core.WorldObj.decode(4, 2, 1), core.Door)
    assert np.asarray(core.Key("yellow")).tolist() == [5, 4, 0] and tuple(core.Key("yellow")) == (5, 4, 0)
    with pytest.raises(ValueError):
        core.Key("magenta")


def test_grid_facade_follows_the_reference_grid():
    """multigrid/core/grid.py:42-195"""
    g = core.Grid(7, 5)
    assert g.state.shape == (7, 5, 3) and (g.state == (1, 0, 0)).all() and g.get(2, 2) is None
    g.wall_rect(0, 0, 7, 5)
    g.horz_wall(2, 2)
    assert (g.state[2:, 2, 0] == 2).all() and g.state[1, 2, 0] == 1
    g.vert_wall(3, 1, 2, obj_type=core.Lava)
    assert g.state[3, 1].tolist() == [9, 0, 0] and isinstance(g.get(3, 1), core.Lava)
    k = core.Key("green")
    g.set(1, 1, k)
    assert g.get(1, 1) is k and g.state[1, 1].tolist() == [5, 1, 0]
    k.color = "red"
    assert g.state[1, 1, 1] == 1                                    # (state is updated by grid.update, grid.py:119-131)
    g.update(1, 1)
    assert g.state[1, 1, 1] == 0
    g.set(1, 1, None)
    assert g.get(1, 1) is None
    with pytest.raises(TypeError):
        g.set(1, 1, (5, 0, 0))


def test_gen_grid_mistakes_are_reported():
    class NoRing(mg.MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = core.Grid(width, height)
            for agent in self.agents:
                agent.state.pos, agent.state.dir = (1, 1), 0

    class Unplaced(mg.MultiGridEnv):
        def _gen_grid(self, width, height):
            self.grid = core.Grid(width, height)
            self.grid.wall_rect(0, 0, width, height)

    class Nothing(mg.MultiGridEnv):
        pass

    kw = dict(grid_size=6, device="cpu", _backend=lambda spec: util.OracleBackend(spec))
    with pytest.raises(ValueError, match="wall"):
        NoRing(**kw).reset(seed=0)                                   # the kernels' precondition (include/mgx.h)
    with pytest.raises(AssertionError):
        Unplaced(**kw).reset(seed=0)                                 # base.py:283-284
    with pytest.raises(NotImplementedError):
        Nothing(**kw).reset(seed=0)
    env = Nothing(**kw)
    with pytest.raises(RuntimeError, match="_gen_grid"):
        env.place_obj(core.Key())
