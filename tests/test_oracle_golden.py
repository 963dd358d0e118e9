"""Pin the CPU oracle (oracle/mgx_oracle.c) against vectors produced by the real reference.

The fixtures in tests/golden/ were written by oracle/gen_golden.py running ini/multigrid itself in the build
container.  Bit-exact on everything: obs images, direction, rewards (float64 equality), terminations,
truncations, post-step grid/agent state, visiting order and the PCG64 stream.
"""
import glob
import json
import os

import numpy as np
import pytest

from oracle import binding as ob

GOLDEN = [p for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
          if not os.path.basename(p).startswith(("layout_", "wrappers_", "custom_", "customsteps_"))]


def load(path):
    z = np.load(path)
    spec = json.loads(str(z["spec_json"]))
    return z, spec


def rng_lohi(words_hilo):
    hi_s, lo_s, hi_i, lo_i = (int(w) for w in words_hilo)
    return np.array([lo_s, hi_s, lo_i, hi_i], dtype=np.uint64)


def test_fixtures_present():
    assert len(GOLDEN) >= 20


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_replays_reference(path):
    z, spec = load(path)
    from tests import util
    env = ob.RefEnv(spec, z["grid0"], z["agents0"], rng_lohi(z["rng0"]), target=[int(v) for v in util.golden_aux(spec)])
    np.testing.assert_array_equal(env.gen_obs(), z["obs0"].astype(np.int64))
    T = z["actions"].shape[0]
    for t in range(T):
        # (*_dictorder fixtures: the reference was stepped with a dict whose keys were inserted in this order)
        obs, direction, reward, terminated, truncated, order = env.step(
            z["actions"][t], hook_order=z["hook_order"][t] if "hook_order" in z.files else None)
        ctx = f"{os.path.basename(path)} step {t}"
        np.testing.assert_array_equal(order, z["order"][t], err_msg=ctx)
        np.testing.assert_array_equal(obs, z["obs"][t].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(direction, z["direction"][t], err_msg=ctx)
        assert reward.tobytes() == z["reward"][t].tobytes(), ctx          # float64 bit equality
        np.testing.assert_array_equal(terminated, z["terminated"][t].astype(bool), err_msg=ctx)
        assert truncated == bool(z["truncated"][t]), ctx
        np.testing.assert_array_equal(env.grid_state, z["grid"][t].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(env.agent_state, z["agents"][t].astype(np.int64), err_msg=ctx)
    np.testing.assert_array_equal(env.rng, rng_lohi(z["rng_final"]))


def test_dict_order_fixtures_pin_the_hook_visiting_order():
    """The *_dictorder fixtures must be cases where the visiting order of the env hooks (the caller's dict order,
    redbluedoors.py:176 / locked_hallway.py:210) changes the result: replayed with ascending order the oracle must DIFFER."""
    from tests import util
    differs = 0
    for path in GOLDEN:
        if "_dictorder_" not in path:
            continue
        z, spec = load(path)
        env = ob.RefEnv(spec, z["grid0"], z["agents0"], rng_lohi(z["rng0"]), target=[int(v) for v in util.golden_aux(spec)])
        for t in range(z["actions"].shape[0]):
            _, _, reward, terminated, _, _ = env.step(z["actions"][t])          # ascending
            if reward.tobytes() != z["reward"][t].tobytes() or not np.array_equal(terminated, z["terminated"][t].astype(bool)):
                differs += 1
                break
    assert differs >= 2


def test_goldens_cover_the_dynamics():
    """The fixture set must actually exercise pickup, drop, door open/close/unlock, box toggle, success,
    failure, truncation and an unseen-masked cell -- otherwise the pin is hollow."""
    seen = dict(pickup=0, drop=0, door_open=0, door_close=0, unlock=0, box_gone=0, success=0, lava=0,
                trunc=0, unseen=0, agent_seen=0)
    for path in GOLDEN:
        z, spec = load(path)
        agents = np.concatenate([z["agents0"][None].astype(np.int64), z["agents"].astype(np.int64)])
        grid = np.concatenate([z["grid0"][None].astype(np.int64), z["grid"].astype(np.int64)])
        carry = agents[:, :, 6]
        seen["pickup"] += int(((carry[:-1] == 1) & (carry[1:] != 1)).sum())
        seen["drop"] += int(((carry[:-1] != 1) & (carry[1:] == 1)).sum())
        door = (grid[:-1, ..., 0] == 4) & (grid[1:, ..., 0] == 4)
        s0, s1 = grid[:-1, ..., 2], grid[1:, ..., 2]
        seen["door_open"] += int((door & (s0 == 1) & (s1 == 0)).sum())
        seen["door_close"] += int((door & (s0 == 0) & (s1 == 1)).sum())
        seen["unlock"] += int((door & (s0 == 2) & (s1 == 0)).sum())
        box_cells = (grid[:-1, ..., 0] == 7) & (grid[1:, ..., 0] == 1)
        picked = ((carry[:-1] == 1) & (carry[1:] == 7)).sum(axis=1)
        seen["box_gone"] += int((box_cells.sum(axis=(1, 2)) - picked > 0).sum())
        seen["success"] += int((z["reward"] > 0).any(axis=1).sum())
        seen["lava"] += int(((z["terminated"].sum(axis=1) > 0) & ~(z["reward"] > 0).any(axis=1)).any())
        seen["trunc"] += int(z["truncated"].any())
        seen["unseen"] += int((z["obs"][..., 0] == 0).any())
        seen["agent_seen"] += int((z["obs"][..., 0] == 10).any())
    missing = [k for k, v in seen.items() if v == 0]
    assert not missing, (missing, seen)


@pytest.mark.parametrize("seed", [0, 7, 123, 2**40 + 5])
def test_pcg64_matches_numpy(seed):
    bg = np.random.PCG64(np.random.SeedSequence(seed))
    st = bg.state["state"]
    m = (1 << 64) - 1
    words = np.array([st["state"] & m, st["state"] >> 64, st["inc"] & m, st["inc"] >> 64], dtype=np.uint64)
    want = np.random.Generator(bg).random(64)
    got = ob.pcg64_random(words, 64)
    assert got.tobytes() == want.tobytes()
    st2 = bg.state["state"]
    assert int(words[0]) | (int(words[1]) << 64) == st2["state"]


def test_unknown_action_raises_value_error():
    path = [p for p in GOLDEN if "empty8_a2_seed0" in p][0]
    z, spec = load(path)
    env = ob.RefEnv(spec, z["grid0"], z["agents0"], rng_lohi(z["rng0"]))
    with pytest.raises(ValueError):
        env.step(np.array([7, 0], dtype=np.int8))


def test_oracle_wrappers_match_reference():
    """OneHotObsWrapper / FullyObsWrapper restatements vs outputs of the real reference wrappers."""
    from tests import util
    assert util.WRAPPER_GOLDEN
    for path in util.WRAPPER_GOLDEN:
        z = np.load(path)
        for t in range(z["obs"].shape[0]):
            np.testing.assert_array_equal(ob.one_hot(z["obs"][t]), z["one_hot"][t])
            np.testing.assert_array_equal(ob.full_obs(z["grid"][t], z["agents"][t]), z["full"][t].astype(np.int64))


PY_GOLDEN = [p for p in GOLDEN if os.path.basename(p).startswith(("empty8_a2_seed0", "empty16_a4_seed1", "empty16_a4_objects", "empty8_a2_unlock",
                                                                  "bup_a2_seed", "emptyrandom6_a3_nooverlap"))]


@pytest.mark.parametrize("path", PY_GOLDEN, ids=[os.path.basename(p)[:-4] for p in PY_GOLDEN])
def test_python_restatement_replays_reference(path):
    """oracle/py_oracle.py -- the pure Python / NumPy per-env restatement bench.py times on one host core as the reference's
    interpreter-speed stand-in -- against what the real reference produced: every output, every step."""
    from oracle import py_oracle as po
    z, spec = load(path)
    assert PY_GOLDEN
    grid, agents = z["grid0"].astype(np.int64), z["agents0"].astype(np.int64)
    hs, ls, hi, li = (int(w) for w in z["rng0"])
    bg = np.random.PCG64()
    st = bg.state
    st["state"] = {"state": (hs << 64) | ls, "inc": (hi << 64) | li}
    st["has_uint32"], st["uinteger"] = 0, 0
    bg.state = st
    rng = np.random.Generator(bg)
    np.testing.assert_array_equal(po.gen_obs(grid, agents, spec["view_size"], spec["see_through_walls"]), z["obs0"])
    sc = 0
    T = min(z["actions"].shape[0], 120)
    for t in range(T):
        obs, d, rew, term, trunc, sc = po.step(spec, grid, agents, rng, sc, z["actions"][t], spec.get("target"))
        ctx = f"step {t}"
        np.testing.assert_array_equal(obs, z["obs"][t], err_msg=ctx)
        np.testing.assert_array_equal(d, z["direction"][t], err_msg=ctx)
        assert rew.tobytes() == z["reward"][t].tobytes(), ctx
        np.testing.assert_array_equal(term.astype(np.uint8), z["terminated"][t], err_msg=ctx)
        assert bool(trunc) == bool(z["truncated"][t]), ctx
        np.testing.assert_array_equal(grid, z["grid"][t].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(agents, z["agents"][t].astype(np.int64), err_msg=ctx)
