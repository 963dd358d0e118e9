"""Fused one-hot output (mgx_gen_obs_one_hot / mgx_step_one_hot): the observation written one-hot encoded by the step's
own launch must equal OneHotObsWrapper.one_hot (multigrid/wrappers.py:158-190) of the plain observation -- checked
against the reference's own wrapper fixtures, against the oracle on random states of many shapes (ragged batches, every
view size, 16-byte-misaligned wave starts), and through the auto-reset step on the BASELINE workloads."""
import json
import zlib

import numpy as np
import pytest
import torch

import multigrid_amd.ops as ops
from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts, workloads
from oracle import binding as ob
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("path", util.WRAPPER_GOLDEN, ids=util.WRAPPER_IDS)
def test_fused_one_hot_vs_reference_wrapper_fixtures(path):
    z = np.load(path)
    spec = EnvSpec.from_dict(json.loads(str(z["spec_json"])))
    T = z["obs"].shape[0]
    env = BatchedMultiGridEnv(spec, T, DEV)
    env.load_state(layouts.grid_to_product(z["grid"]), layouts.pack_agents(z["agents"]), validate=False)
    oh, dirs = env.gen_obs(one_hot=True)
    np.testing.assert_array_equal(oh.cpu().numpy(), z["one_hot"])


CASES = [
    ("C2_empty16_a4_v7", EnvSpec(16, 16, 4, 7, max_steps=1024), 4096, 6),
    ("bup_11x6_a2", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 3001, 6),
    ("C5_64x64_a16_v9", EnvSpec(64, 64, 16, 9, max_steps=16384), 130, 3),
    ("ragged_a3_v5", EnvSpec(9, 7, 3, 5, max_steps=50, allow_agent_overlap=False), 1001, 5),
    ("a1_v3_seethrough", EnvSpec(8, 8, 1, 3, max_steps=30, see_through_walls=True), 777, 4),
    ("a5_v11", EnvSpec(13, 12, 5, 11, max_steps=40), 333, 3),
    ("a7_v13", EnvSpec(20, 17, 7, 13, max_steps=40), 129, 3),
    ("a2_v15", EnvSpec(24, 24, 2, 15, max_steps=40), 65, 3),
    ("a32_v7", EnvSpec(12, 12, 32, 7, max_steps=40), 37, 3),
    ("single_env", EnvSpec(8, 8, 2, 7, max_steps=256), 1, 5),
]


@pytest.mark.parametrize("name,spec,B,T", CASES, ids=[c[0] for c in CASES])
def test_fused_one_hot_step_vs_oracle(name, spec, B, T):
    st = util.random_state(spec, B, seed=zlib.crc32(name.encode()) % 10000)
    env = BatchedMultiGridEnv(spec, B, DEV)
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    ref = {k: v.copy() for k, v in st.items()}
    sd = spec.as_dict()
    o_ref, d_ref = ob.gen_obs_batch(sd, ref["grid"], ref["agents"], nthreads=8)
    oh, dirs = env.gen_obs(one_hot=True)
    assert oh.shape == (B, spec.num_agents, spec.view_size, spec.view_size, 21)
    np.testing.assert_array_equal(oh.cpu().numpy(), ob.one_hot(o_ref))
    np.testing.assert_array_equal(dirs.cpu().numpy(), d_ref)
    for t in range(T):
        act = util.random_actions(B, spec.num_agents, seed=1000 + t)
        o_ref, d_ref, r_ref, te_ref, tr_ref = ob.step_batch(
            sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["target"], nthreads=8)
        oh, dirs, rew, term, trunc = env.step(torch.from_numpy(act).to(DEV), one_hot=True)
        ctx = f"{name} step {t}"
        assert oh.cpu().numpy().tobytes() == ob.one_hot(o_ref).tobytes(), ctx
        np.testing.assert_array_equal(dirs.cpu().numpy(), d_ref, err_msg=ctx)
        assert rew.cpu().numpy().tobytes() == r_ref.tobytes(), ctx
        np.testing.assert_array_equal(term.cpu().numpy(), te_ref, err_msg=ctx)
        np.testing.assert_array_equal(trunc.cpu().numpy(), tr_ref, err_msg=ctx)
        np.testing.assert_array_equal(env.grid.cpu().numpy(), ref["grid"], err_msg=ctx)
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"], err_msg=ctx)
    env.check_errors()


@pytest.mark.parametrize("name,B", [("c2", 4096), ("c3", 5000), ("c5", 300)])
def test_fused_one_hot_autoreset_equals_step_then_one_hot(name, B):
    """mgx_step_one_hot with auto-reset == mgx_step_autoreset followed by mgx_one_hot, on every output and the state."""
    wl = workloads.make(name, batch=B, cell_bytes=2)           # (one-hot output: 16-bit cells; C5's own format is the compact one)
    a, b = wl.make_env(DEV), wl.make_env(DEV)
    if name != "c5":
        a.step_count.fill_(wl.spec.max_steps - 3); b.step_count.fill_(wl.spec.max_steps - 3)      # restarts on the way
    g = torch.Generator(device=DEV); g.manual_seed(4)
    for t in range(8):
        act = torch.randint(0, 7, (B, wl.spec.num_agents), dtype=torch.int8, device=DEV, generator=g)
        want = [x.clone() for x in a.step(act, auto_reset=True)]
        want_oh = a.one_hot_obs().clone()
        got = b.step(act, auto_reset=True, one_hot=True)
        assert torch.equal(got[0], want_oh), f"step {t}"
        for x, y in zip(got[1:], want[1:]):
            assert torch.equal(x, y), f"step {t}"
        assert torch.equal(a.was_reset, b.was_reset)
    for f in ("grid", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert name == "c5" or int(a.episode.sum()) >= B


def test_one_hot_torch_ops():
    spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
    B = 515
    st = util.random_state(spec, B, seed=12)
    ints = ops.spec_to_ints(spec)
    g = util.dev_cells(st["grid"], DEV); a = torch.from_numpy(st["agents"]).to(DEV)
    oh, dirs = torch.ops.mgx.gen_obs_one_hot(g, a, ints)
    o_ref, d_ref = ob.gen_obs_batch(spec.as_dict(), st["grid"], st["agents"])
    np.testing.assert_array_equal(oh.cpu().numpy(), ob.one_hot(o_ref))
    rng = torch.from_numpy(st["rng"].view(np.int64)).to(DEV); sc = torch.from_numpy(st["step_count"]).to(DEV)
    err = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=DEV)
    ref = {k: v.copy() for k, v in st.items()}
    act = util.random_actions(B, 4, seed=3)
    want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, None, nthreads=8)
    got = torch.ops.mgx.step_one_hot(g, a, rng, sc, torch.from_numpy(act).to(DEV), None, err, None, None, None, None, 0, ints)
    assert got[0].cpu().numpy().tobytes() == ob.one_hot(want[0]).tobytes()
    for x, w in zip(got[1:5], want[1:]):
        assert x.cpu().numpy().tobytes() == w.tobytes()
    assert util.grid3(g).tobytes() == ref["grid"].tobytes()


@pytest.mark.parametrize("kind", ["bup", "empty_random"])
def test_one_hot_with_device_generation_in_one_launch(kind):
    """OH x GEN (the production combination: RLlib's default registration one-hots every env, rllib/__init__.py:110-111, and
    episodes restart through reset, base.py:250-301): the step with one-hot output AND the regeneration of the envs it finished,
    in ONE launch == mgx_step_generate followed by mgx_one_hot of its observation; every output, the whole state, the generators."""
    if kind == "bup":
        wl = workloads.make("c3", batch=3000, first_env=0, global_batch=16384)
        gen = dict(kind="blockedunlockpickup", room_size=6)
    else:
        wl = workloads.make("c2", batch=2500, first_env=64, global_batch=4096)
        gen = dict(kind="empty_random")
    B, A = wl.batch, wl.spec.num_agents

    def mk():
        e = wl.make_env(DEV, auto_reset=False)
        e.set_layout_generator(gen["kind"], layout_seed=11, room_size=gen.get("room_size", 0))
        e.step_count.fill_(wl.spec.max_steps - 3)                         # every env is regenerated within the run
        return e
    a, b = mk(), mk()
    g = torch.Generator(device=DEV); g.manual_seed(9)
    for t in range(8):
        act = torch.randint(0, 7, (B, A), dtype=torch.int8, device=DEV, generator=g)
        want = [x.clone() for x in a.step(act, auto_reset=True)]
        want_oh = a.one_hot_obs().clone()
        got = b.step(act, auto_reset=True, one_hot=True)
        assert torch.equal(got[0], want_oh), f"{kind} step {t}: one-hot observation"
        for x, y in zip(got[1:], want[1:]):
            assert torch.equal(x, y), f"{kind} step {t}"
        assert torch.equal(a.was_reset, b.was_reset)
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert torch.equal(a._gen["gen_state"], b._gen["gen_state"]) and int(b.episode.sum()) >= B
    b.check_errors()


@pytest.mark.parametrize("name,B,T", [("c2", 1000, 9), ("c3", 2048, 7), ("c5", 100, 4)])
def test_rollout_with_one_hot_output_equals_steps(name, B, T):
    """mgx_rollout with one-hot observations (T steps, one launch, u8[T,B,A,v,v,21]) == T x mgx_step_one_hot, with the fused
    auto-reset on the way; also through torch.ops.mgx.rollout_one_hot vs the oracle."""
    wl = workloads.make(name, batch=B, cell_bytes=2)           # (rollouts and one-hot output: 16-bit cells)
    a, b = wl.make_env(DEV), wl.make_env(DEV)
    if name != "c5":
        a.step_count.fill_(wl.spec.max_steps - 3); b.step_count.fill_(wl.spec.max_steps - 3)
    g = torch.Generator(device=DEV); g.manual_seed(5)
    acts = torch.randint(0, 7, (T, B, wl.spec.num_agents), dtype=torch.int8, device=DEV, generator=g)
    out = b.rollout(acts, auto_reset=True, one_hot=True)
    assert tuple(out["obs"].shape) == (T, B, wl.spec.num_agents, wl.spec.view_size, wl.spec.view_size, 21)
    for t in range(T):
        want = [x.clone() for x in a.step(acts[t], auto_reset=True, one_hot=True)]
        for k, key in enumerate(("obs", "dir", "reward", "terminated", "truncated")):
            assert torch.equal(out[key][t], want[k]), f"{name} step {t}: {key}"
        assert torch.equal(out["was_reset"][t], a.was_reset)
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    if name == "c2":
        spec = wl.spec
        st = util.random_state(spec, 300, seed=21)
        ints = ops.spec_to_ints(spec)
        gd = util.dev_cells(st["grid"], DEV); ad = torch.from_numpy(st["agents"]).to(DEV)
        rng = torch.from_numpy(st["rng"].view(np.int64)).to(DEV); sc = torch.from_numpy(st["step_count"]).to(DEV)
        err = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=DEV)
        act3 = np.stack([util.random_actions(300, 4, seed=70 + t) for t in range(3)])
        got = torch.ops.mgx.rollout_one_hot(gd, ad, rng, sc, torch.from_numpy(act3).to(DEV), None, err, ints)
        ref = {k: v.copy() for k, v in st.items()}
        for t in range(3):
            w = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act3[t], None, nthreads=8)
            assert got[0][t].cpu().numpy().tobytes() == ob.one_hot(w[0]).tobytes(), t
