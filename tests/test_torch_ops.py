"""The torch.ops.mgx.* surface (SURVEY.md section 8b "must export"): every op called through the dispatcher on HIP
tensors and compared with the oracle, including the in-place schema arguments; CPU tensors are refused."""
import numpy as np
import pytest
import torch

import multigrid_amd.ops as ops
from multigrid_amd import EnvSpec, layouts, workloads
from oracle import binding as ob
from tests import util
from tests.test_full_size import oracle_reset_done

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev_state(st, spec):
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in st.items() if v is not None}
    t["grid"] = util.dev_cells(st["grid"], DEV)                      # the ops take packed cells (include/mgx.h MgxCell)
    t["rng"] = torch.from_numpy(st["rng"].view(np.int64)).to(DEV)
    t["err"] = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=DEV)
    return t


@pytest.mark.parametrize("kind", ["empty", "blockedunlockpickup"])
def test_step_op_vs_oracle_in_place_args(kind):
    spec = (EnvSpec(16, 16, 4, 7, max_steps=1024) if kind == "empty"
            else EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind=kind))
    B, ints = 777, None
    st = util.random_state(spec, B, seed=31)
    st = dict(grid=st["grid"], agents=st["agents"], rng=st["rng"], step_count=st["step_count"],
              aux=st["target"] if kind != "empty" else None)
    ref = {k: (v.copy() if v is not None else None) for k, v in st.items()}
    d = _dev_state(st, spec)
    ints = ops.spec_to_ints(spec)
    for t in range(12):
        act = util.random_actions(B, spec.num_agents, seed=70 + t)
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["aux"], nthreads=8)
        got = torch.ops.mgx.step(d["grid"], d["agents"], d["rng"], d["step_count"], torch.from_numpy(act).to(DEV),
                                 d.get("aux"), d["err"], ints)
        assert len(got) == 5
        for g, w in zip(got, want):
            assert g.cpu().numpy().tobytes() == w.tobytes(), f"step {t}"
        # the (a!)..(d!) arguments were updated in place
        assert util.grid3(d["grid"]).tobytes() == ref["grid"].tobytes()
        assert d["agents"].cpu().numpy().tobytes() == ref["agents"].tobytes()
        np.testing.assert_array_equal(d["rng"].cpu().numpy().view(np.uint64), ref["rng"])
        np.testing.assert_array_equal(d["step_count"].cpu().numpy(), ref["step_count"])
    assert d["err"].cpu().tolist() == [0, 2 ** 31 - 1]
    bad = torch.full((B, spec.num_agents), 9, dtype=torch.int8, device=DEV)
    torch.ops.mgx.step(d["grid"], d["agents"], d["rng"], d["step_count"], bad, d.get("aux"), d["err"], ints)
    assert int(d["err"][0]) > 0                                     # (e!): unknown actions are counted (base.py:473-474)


def test_step_op_refuses_cpu_tensors_and_bad_shapes():
    spec = EnvSpec(8, 8, 2, 7, max_steps=64)
    st = util.random_state(spec, 8, seed=1)
    c = dict(grid=torch.from_numpy(layouts.pack_cells(st["grid"]).view(np.int16)), agents=torch.from_numpy(st["agents"]),
             rng=torch.from_numpy(st["rng"].view(np.int64)), sc=torch.from_numpy(st["step_count"]),
             act=torch.zeros((8, 2), dtype=torch.int8), err=torch.zeros(2, dtype=torch.int32))
    ints = ops.spec_to_ints(spec)
    with pytest.raises(NotImplementedError):                         # no CPU dispatch key: there is no CPU product path
        torch.ops.mgx.step(c["grid"], c["agents"], c["rng"], c["sc"], c["act"], None, c["err"], ints)
    g = {k: v.to(DEV) for k, v in c.items()}
    with pytest.raises((ValueError, RuntimeError)):
        torch.ops.mgx.step(g["grid"], g["agents"], g["rng"], g["sc"], g["act"][:, :1].contiguous(), None, g["err"], ints)
    with pytest.raises((TypeError, RuntimeError)):
        torch.ops.mgx.step(g["grid"], g["agents"], g["rng"], g["sc"], g["act"].to(torch.int32), None, g["err"], ints)


def test_step_autoreset_op_vs_oracle():
    wl = workloads.make("c3", batch=2000)
    wl.agents[::2, 0, 1:4] = wl.agents[::2, 0, 1:4]                  # (layout pool starts as they are)
    spec, B, A = wl.spec, wl.batch, wl.spec.num_agents
    ref = dict(grid=wl.grid.copy(), agents=wl.agents.copy(), rng=wl.rng.copy(), step_count=np.zeros(B, np.int32), aux=wl.aux.copy())
    # make restarts happen: half the envs are one step from truncation
    ref["step_count"][::2] = spec.max_steps - 2
    d = _dev_state(ref, spec)
    pool = [util.dev_cells(wl.pool[0], DEV)] + [torch.from_numpy(p).to(DEV) for p in wl.pool[1:]]
    episode_ref = np.zeros(B, np.int32)
    episode = torch.zeros(B, dtype=torch.int32, device=DEV)
    ints = ops.spec_to_ints(spec)
    resets = 0
    for t in range(8):
        act = util.random_actions(B, A, seed=t, p_missing=0.0)
        was_ref = oracle_reset_done(wl, ref, episode_ref)
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["aux"], nthreads=8)
        got = torch.ops.mgx.step_autoreset(d["grid"], d["agents"], d["rng"], d["step_count"], torch.from_numpy(act).to(DEV),
                                           d["aux"], d["err"], pool[0], pool[1], pool[2], episode, wl.first_env, ints)
        assert len(got) == 6
        for g, w in zip(got[:5], want):
            assert g.cpu().numpy().tobytes() == w.tobytes(), f"step {t}"
        np.testing.assert_array_equal(got[5].cpu().numpy(), was_ref)
        np.testing.assert_array_equal(episode.cpu().numpy(), episode_ref)
        assert util.grid3(d["grid"]).tobytes() == ref["grid"].tobytes()
        np.testing.assert_array_equal(d["aux"].cpu().numpy(), ref["aux"])
        resets += int(was_ref.sum())
    assert resets >= B // 2


def test_rollout_op_equals_step_op():
    spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
    B, T = 1000, 10
    st = util.random_state(spec, B, seed=8)
    st = dict(grid=st["grid"], agents=st["agents"], rng=st["rng"], step_count=st["step_count"], aux=None)
    a, b = _dev_state(st, spec), _dev_state(st, spec)
    acts = torch.from_numpy(np.stack([util.random_actions(B, 4, seed=t) for t in range(T)])).to(DEV)
    ints = ops.spec_to_ints(spec)
    out = torch.ops.mgx.rollout(a["grid"], a["agents"], a["rng"], a["step_count"], acts, None, a["err"], ints)
    for t in range(T):
        got = torch.ops.mgx.step(b["grid"], b["agents"], b["rng"], b["step_count"], acts[t], None, b["err"], ints)
        for x, y in zip(out, got):
            assert torch.equal(x[t], y), f"step {t}"
    for k in ("grid", "agents", "rng", "step_count"):
        assert torch.equal(a[k], b[k]), k


def test_ops_accept_the_reference_byte_grids():
    """SURVEY.md section 8b words the op signatures on `grid_u8[B,H,W,3]` -- the reference's (type, color, state) triples.  The
    ops take that form too (packed on the way in, unpacked into the caller's tensor on the way out): same results as the
    packed form and as the oracle, in-place arguments updated.  No host synchronisation on that path: a value the packed cells
    cannot hold is stored truncated, and torch.ops.mgx.pack_grid reports the count for callers who want to check."""
    spec = EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup")
    B, ints = 333, None
    st = util.random_state(spec, B, seed=17)
    st = dict(grid=st["grid"], agents=st["agents"], rng=st["rng"], step_count=st["step_count"], aux=st["target"])
    ref = {k: v.copy() for k, v in st.items()}
    d = _dev_state(st, spec)
    d["grid"] = torch.from_numpy(st["grid"]).to(DEV)                   # bytes, not packed cells
    ints = ops.spec_to_ints(spec)
    obs, dirs = torch.ops.mgx.gen_obs(d["grid"], d["agents"], ints)
    o_ref, d_ref = ob.gen_obs_batch(spec.as_dict(), ref["grid"], ref["agents"])
    assert obs.cpu().numpy().tobytes() == o_ref.tobytes() and dirs.cpu().numpy().tobytes() == d_ref.tobytes()
    full = torch.ops.mgx.full_obs(d["grid"], d["agents"], ints)
    assert torch.equal(full, torch.ops.mgx.full_obs(util.dev_cells(st["grid"], DEV), d["agents"], ints))
    for t in range(6):
        act = util.random_actions(B, 2, seed=40 + t)
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["aux"], nthreads=8)
        got = torch.ops.mgx.step(d["grid"], d["agents"], d["rng"], d["step_count"], torch.from_numpy(act).to(DEV), d["aux"], d["err"], ints)
        for g, w in zip(got, want):
            assert g.cpu().numpy().tobytes() == w.tobytes(), f"step {t}"
        assert d["grid"].dtype == torch.uint8 and d["grid"].cpu().numpy().tobytes() == ref["grid"].tobytes()      # (a!) as bytes
    acts = torch.from_numpy(np.stack([util.random_actions(B, 2, seed=60 + t) for t in range(4)])).to(DEV)
    out = torch.ops.mgx.rollout(d["grid"], d["agents"], d["rng"], d["step_count"], acts, d["aux"], d["err"], ints)
    for t in range(4):
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], acts[t].cpu().numpy(), ref["aux"], nthreads=8)
        assert out[0][t].cpu().numpy().tobytes() == want[0].tobytes()
    assert d["grid"].cpu().numpy().tobytes() == ref["grid"].tobytes()
    bad = d["grid"].clone(); bad[0, 2, 2, 0] = 99                                                                  # no such type
    cells, n_bad = torch.ops.mgx.pack_grid(bad)
    assert int(n_bad[0]) == 1 and int(torch.ops.mgx.pack_grid(d["grid"])[1][0]) == 0
    torch.ops.mgx.gen_obs(bad, d["agents"], ints)                                                                  # (no sync, no raise)


def test_ops_are_the_compiled_library_and_step_ordered_follows_the_dict_order():
    """torch.ops.mgx.* come from lib/libmgx_torch.so (TORCH_LIBRARY, csrc/mgx_torch.cpp), not from Python registrations; the
    hook-order form of the step (redbluedoors.py:176: `for agent_id, action in actions.items()`) against the reference's own
    fixture recorded with a reversed dict."""
    import os
    from multigrid_amd import layouts
    assert os.path.basename(ops.TORCH_LIB_PATH) == "libmgx_torch.so" and int(torch.ops.mgx.abi_version()) == 8
    with open(f"/proc/{os.getpid()}/maps") as fh:
        assert "libmgx_torch.so" in fh.read()
    z, d_, spec = util.load_golden([p for p in util.GOLDEN if "rbd_a3_dictorder_rev" in p][0])
    ints = ops.spec_to_ints(spec)
    grid = util.dev_cells(layouts.grid_to_product(z["grid0"])[None], DEV)
    agents = torch.from_numpy(layouts.pack_agents(z["agents0"])[None]).to(DEV)
    rng = torch.from_numpy(util.rng_words_lohi(z["rng0"]).view(np.int64)[None].copy()).to(DEV)
    sc = torch.zeros(1, dtype=torch.int32, device=DEV)
    aux = torch.from_numpy(util.golden_aux(d_)[None]).to(DEV)
    err = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=DEV)
    for t in range(z["actions"].shape[0]):
        out = torch.ops.mgx.step_ordered(grid, agents, rng, sc, torch.from_numpy(z["actions"][t][None]).to(DEV),
                                         torch.from_numpy(z["hook_order"][t][None]).to(DEV), aux, err, ints)
        np.testing.assert_array_equal(out[0][0].cpu().numpy(), z["obs"][t])
        assert out[2][0].cpu().numpy().tobytes() == z["reward"][t].tobytes()
        np.testing.assert_array_equal(out[3][0].cpu().numpy(), z["terminated"][t])
    assert int(err[0]) == 0
