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
    torch.ops.mgx.gen_obs(bad, d["agents"], ints)                                                                  # (no sync, no raise ...
    with pytest.raises(RuntimeError, match="cannot hold"):                           # ... the report is deferred: the next op, or:)
        torch.ops.mgx.check_errors(0)
    torch.ops.mgx.check_errors(0)                                                    # (reported once)


def test_out_variants_allocate_nothing_and_equal_the_allocating_ops():
    """torch.ops.mgx.step_out / step_autoreset_out / step_one_hot_out / gen_obs_out: the outputs go into the caller's tensors
    (multigrid/base.py:303-346 per call without five allocations); same bytes as the allocating forms."""
    wl = workloads.make("c2", batch=1024, global_batch=1024)
    spec = wl.spec
    ints = ops.spec_to_ints(spec)
    e1, e2 = wl.make_env(torch.device(DEV), auto_reset=True), wl.make_env(torch.device(DEV), auto_reset=True)
    for e in (e1, e2):
        e.step_count.fill_(spec.max_steps - 3)                       # (truncation resets inside the run)
    B, A, v = 1024, spec.num_agents, spec.view_size
    o = dict(obs=torch.zeros((B, A, v, v, 3), dtype=torch.uint8, device=DEV), dir=torch.zeros((B, A), dtype=torch.uint8, device=DEV),
             rew=torch.zeros((B, A), dtype=torch.float64, device=DEV), term=torch.zeros((B, A), dtype=torch.uint8, device=DEV),
             trunc=torch.zeros((B,), dtype=torch.uint8, device=DEV), was=torch.zeros((B,), dtype=torch.uint8, device=DEV))
    oh = torch.zeros((B, A, v, v, 21), dtype=torch.uint8, device=DEV)
    pg, pa, _ = e1._pool
    for t in range(8):
        act = torch.from_numpy(util.random_actions(B, A, seed=t)).to(DEV)
        want = torch.ops.mgx.step_autoreset(e1.cells, e1.agents, e1.rng, e1.step_count, act, None, e1.err, pg, pa, None, e1.episode, 0, ints)
        before = torch.cuda.memory_allocated()
        ret = torch.ops.mgx.step_autoreset_out(e2.cells, e2.agents, e2.rng, e2.step_count, act, None, e2.err, pg, pa, None, e2.episode, 0,
                                               ints, o["obs"], o["dir"], o["rew"], o["term"], o["trunc"], o["was"])
        assert ret is None and torch.cuda.memory_allocated() == before
        for w, g in zip(want, (o["obs"], o["dir"], o["rew"], o["term"], o["trunc"], o["was"])):
            assert torch.equal(w, g), t
    assert torch.equal(e1.cells, e2.cells) and int(e1.episode.sum()) >= B
    act = torch.from_numpy(util.random_actions(B, A, seed=99)).to(DEV)
    want = torch.ops.mgx.step(e1.cells, e1.agents, e1.rng, e1.step_count, act, None, e1.err, ints)
    torch.ops.mgx.step_out(e2.cells, e2.agents, e2.rng, e2.step_count, act, None, e2.err, ints, o["obs"], o["dir"], o["rew"], o["term"], o["trunc"])
    for w, g in zip(want, (o["obs"], o["dir"], o["rew"], o["term"], o["trunc"])):
        assert torch.equal(w, g)
    want = torch.ops.mgx.step_one_hot(e1.cells, e1.agents, e1.rng, e1.step_count, act, None, e1.err, None, None, None, None, 0, ints)
    torch.ops.mgx.step_one_hot_out(e2.cells, e2.agents, e2.rng, e2.step_count, act, None, e2.err, None, None, None, None, 0, ints,
                                   oh, o["dir"], o["rew"], o["term"], o["trunc"], None)
    assert torch.equal(want[0], oh) and torch.equal(want[2], o["rew"])
    w_obs, w_dir = torch.ops.mgx.gen_obs(e1.cells, e1.agents, ints)
    torch.ops.mgx.gen_obs_out(e2.cells, e2.agents, ints, o["obs"], o["dir"])
    assert torch.equal(w_obs, o["obs"]) and torch.equal(w_dir, o["dir"])
    w_oh, _ = torch.ops.mgx.gen_obs_one_hot(e1.cells, e1.agents, ints)
    torch.ops.mgx.gen_obs_out(e2.cells, e2.agents, ints, oh, o["dir"])
    assert torch.equal(w_oh, oh)
    with pytest.raises(ValueError, match="obs"):                     # wrong shape of an output
        torch.ops.mgx.step_out(e2.cells, e2.agents, e2.rng, e2.step_count, act, None, e2.err, ints, oh, o["dir"], o["rew"], o["term"], o["trunc"])
    with pytest.raises(TypeError, match="reward"):
        torch.ops.mgx.step_out(e2.cells, e2.agents, e2.rng, e2.step_count, act, None, e2.err, ints, o["obs"], o["dir"], o["rew"].float(),
                               o["term"], o["trunc"])


def test_bound_step_equals_the_out_variant_and_checks_only_the_actions():
    """torch.ops.mgx.bind_step / step_bound / unbind_step: the pointers of a step resolved and checked once, then (handle, actions)
    per call -- for a caller that steps the same tensors in a closed loop (multigrid/base.py:303-346 called per step;
    multigrid/rllib/__init__.py:59-63).  Same bytes as step_autoreset_out, with the fused auto-reset and the one-hot form."""
    wl = workloads.make("c2", batch=1024, global_batch=1024)
    spec = wl.spec
    ints = ops.spec_to_ints(spec)
    e1, e2 = wl.make_env(torch.device(DEV), auto_reset=True), wl.make_env(torch.device(DEV), auto_reset=True)
    for e in (e1, e2):
        e.step_count.fill_(spec.max_steps - 3)
    B, A, v = 1024, spec.num_agents, spec.view_size

    def outs(last):
        return [torch.zeros((B, A, v, v, last), dtype=torch.uint8, device=DEV), torch.zeros((B, A), dtype=torch.uint8, device=DEV),
                torch.zeros((B, A), dtype=torch.float64, device=DEV), torch.zeros((B, A), dtype=torch.uint8, device=DEV),
                torch.zeros((B,), dtype=torch.uint8, device=DEV), torch.zeros((B,), dtype=torch.uint8, device=DEV)]
    o1, o2 = outs(3), outs(3)
    pg, pa, _ = e1._pool
    h = torch.ops.mgx.bind_step(e2.cells, e2.agents, e2.rng, e2.step_count, None, e2.err, ints, o2[0], o2[1], o2[2], o2[3], o2[4],
                                pg, pa, None, e2.episode, 0, o2[5], False)
    assert isinstance(h, int) and h >= 0
    for t in range(8):
        act = torch.from_numpy(util.random_actions(B, A, seed=t)).to(DEV)
        torch.ops.mgx.step_autoreset_out(e1.cells, e1.agents, e1.rng, e1.step_count, act, None, e1.err, pg, pa, None, e1.episode, 0,
                                         ints, *o1)
        before = torch.cuda.memory_allocated()
        assert torch.ops.mgx.step_bound(h, act) is None and torch.cuda.memory_allocated() == before
        for w, g in zip(o1, o2):
            assert torch.equal(w, g), t
    assert torch.equal(e1.cells, e2.cells) and torch.equal(e1.agents, e2.agents) and torch.equal(e1.rng, e2.rng)
    assert torch.equal(e1.episode, e2.episode) and int(e1.episode.sum()) >= B
    with pytest.raises(ValueError, match="actions"):
        torch.ops.mgx.step_bound(h, torch.zeros((B, A + 1), dtype=torch.int8, device=DEV))
    with pytest.raises(TypeError, match="actions"):
        torch.ops.mgx.step_bound(h, torch.zeros((B, A), dtype=torch.int32, device=DEV))
    torch.ops.mgx.unbind_step(h)
    with pytest.raises(ValueError, match="no such handle"):
        torch.ops.mgx.step_bound(h, torch.zeros((B, A), dtype=torch.int8, device=DEV))
    # the one-hot form without auto-reset; a second handle reuses the freed slot
    oh1, oh2 = outs(21), outs(21)
    h2 = torch.ops.mgx.bind_step(e2.cells, e2.agents, e2.rng, e2.step_count, None, e2.err, ints, oh2[0], oh2[1], oh2[2], oh2[3], oh2[4],
                                 None, None, None, None, 0, None, True)
    assert h2 != h and (h2 & 0xFFFFF) == (h & 0xFFFFF)          # the same slot under a new generation: ...
    with pytest.raises(ValueError, match="no such handle"):        # ... the stale handle does not step what now lives there
        torch.ops.mgx.step_bound(h, torch.zeros((B, A), dtype=torch.int8, device=DEV))
    act = torch.from_numpy(util.random_actions(B, A, seed=77)).to(DEV)
    torch.ops.mgx.step_one_hot_out(e1.cells, e1.agents, e1.rng, e1.step_count, act, None, e1.err, None, None, None, None, 0, ints,
                                   oh1[0], oh1[1], oh1[2], oh1[3], oh1[4], None)
    torch.ops.mgx.step_bound(h2, act)
    for w, g in zip(oh1[:5], oh2[:5]):
        assert torch.equal(w, g)
    torch.ops.mgx.unbind_step(h2)
    with pytest.raises(TypeError, match="packed grid"):               # byte grids are converted per call: the out-variants take them
        torch.ops.mgx.bind_step(torch.zeros((B, 16, 16, 3), dtype=torch.uint8, device=DEV), e2.agents, e2.rng, e2.step_count, None, e2.err,
                                ints, o2[0], o2[1], o2[2], o2[3], o2[4], None, None, None, None, 0, None, False)


def test_the_wall_ring_contract_is_enforced_at_the_op_boundary():
    """include/mgx.h: the outer ring of every env's grid is the reference's WALL (what multigrid/utils/obs.py:199-202 shows for
    cells outside the grid; every _gen_grid starts from Grid.wall_rect).  A byte grid that breaks it is reported -- deferred, without
    a host synchronisation on the call itself; packed state is checked on request by torch.ops.mgx.check_grid."""
    spec = EnvSpec(9, 7, 2, 5, max_steps=50)
    B = 64
    st = util.random_state(spec, B, seed=5)
    ints = ops.spec_to_ints(spec)
    agents = torch.from_numpy(st["agents"]).to(DEV)
    good = torch.from_numpy(st["grid"]).to(DEV)
    torch.ops.mgx.gen_obs(good, agents, ints)
    torch.ops.mgx.check_errors(0)                                    # a valid grid: nothing to report
    holed = good.clone(); holed[3, 0, 4] = torch.tensor([1, 0, 0], dtype=torch.uint8)       # a hole in env 3's top wall
    holed[5, 6, 0] = torch.tensor([2, 1, 0], dtype=torch.uint8)                              # a GREEN wall in env 5's ring
    torch.ops.mgx.gen_obs(holed, agents, ints)                       # (returns: the launch is asynchronous)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="2 outer-ring cell"):
        torch.ops.mgx.gen_obs(good, agents, ints)                    # the next op on the device raises
    torch.ops.mgx.gen_obs(good, agents, ints)                        # (once)
    # the plain step takes a byte grid AS IT IS (MgxSpec.cell_bytes = 3: packed by the kernel's own load phase, which counts the same
    # things) -- the report is queued with the first such call and by check_errors(), which waits
    rng = torch.from_numpy(st["rng"].view(np.int64)).to(DEV); sc = torch.from_numpy(st["step_count"]).to(DEV)
    err = torch.tensor([0, 2 ** 31 - 1], dtype=torch.int32, device=DEV)
    act = torch.zeros((B, 2), dtype=torch.int8, device=DEV)
    torch.ops.mgx.step(good.clone(), agents.clone(), rng.clone(), sc.clone(), act, None, err, ints)
    torch.ops.mgx.check_errors(0)
    for _ in range(3):                                               # (only the first call queues a report by itself)
        torch.ops.mgx.step(good.clone(), agents.clone(), rng.clone(), sc.clone(), act, None, err, ints)
    torch.ops.mgx.step(holed.clone(), agents.clone(), rng.clone(), sc.clone(), act, None, err, ints)
    with pytest.raises(RuntimeError, match="2 outer-ring cell"):
        torch.ops.mgx.check_errors(0)
    torch.ops.mgx.check_errors(0)
    # packed state, on request
    cells = util.dev_cells(st["grid"], DEV)
    assert torch.ops.mgx.check_grid(cells, agents, ints).cpu().tolist() == [0, 0, 0, 2 ** 31 - 1]
    bad = cells.clone()
    bad[7, 0, 0] = 0x0001                                            # ring cell = empty
    bad[9, 3, 3] = 0x0502                                            # a wall without its opaque bit
    bad[11, 2, 2] = 0x00f1                                           # reserved bits
    ag = agents.clone(); ag[20, 1, 2] = 0; ag[21, 0, 1] = 7          # an agent on the ring; a direction of 7
    assert torch.ops.mgx.check_grid(bad, ag, ints).cpu().tolist() == [2, 1, 2, 7]
    # a tensor of another kind of device is refused before anything is launched
    with pytest.raises(RuntimeError):
        torch.ops.mgx.check_grid(cells, agents.cpu(), ints)


def test_ops_are_the_compiled_library_and_step_ordered_follows_the_dict_order():
    """torch.ops.mgx.* come from lib/libmgx_torch.so (TORCH_LIBRARY, csrc/mgx_torch.cpp), not from Python registrations; the
    hook-order form of the step (redbluedoors.py:176: `for agent_id, action in actions.items()`) against the reference's own
    fixture recorded with a reversed dict."""
    import os
    from multigrid_amd import layouts
    assert os.path.basename(ops.TORCH_LIB_PATH) == "libmgx_torch.so" and int(torch.ops.mgx.abi_version()) == 11
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
