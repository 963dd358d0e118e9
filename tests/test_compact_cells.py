"""COMPACT one-byte grid cells (include/mgx.h: MgxCell8, EnvSpec.cell_bytes = 1 -- the format for large grids: type and state
coded jointly, a third less traffic per step and twice the envs per wavefront at 64x64) against the oracle, which knows nothing
about cell formats, and against the 16-bit format: the same results bit for bit.

`-m gpu`: the HIP kernels through the C ABI (random states on every view size incl. the packed remainders of 9x9 views, the
throughput / streamed families, hooks, the fused auto-reset, sub-shard chains, full_obs, check_grid, pack / unpack) and C5 at its
full size in BOTH formats.  `-m "not gpu"`: the host side of the format (packing, BatchedMultiGridEnv on the oracle backend, what is
refused)."""
import dataclasses
import zlib

import numpy as np
import pytest
import torch

from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts, workloads
from oracle import binding as ob
from tests import util

DEV = "cuda:0"


def compact(spec: EnvSpec) -> EnvSpec:
    return dataclasses.replace(spec, cell_bytes=1)


# ---------------------------------------------------------------------------------------------------- CPU: host side of the format
def test_env_on_the_oracle_backend_holds_compact_cells_and_steps_like_the_wide_one():
    spec = EnvSpec(12, 10, 3, 7, max_steps=50)
    B = 16
    st = util.random_state(spec, B, seed=5)
    envs = []
    for sp in (spec, compact(spec)):
        env = BatchedMultiGridEnv(sp, B, "cpu", backend=util.OracleBackend(sp))
        env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
        envs.append(env)
    wide, comp = envs
    assert wide.cells.dtype == torch.int16 and comp.cells.dtype == torch.uint8
    np.testing.assert_array_equal(comp.cells.numpy(), layouts.pack_cells8(st["grid"]))
    np.testing.assert_array_equal(comp.grid.numpy(), st["grid"])                  # the unpacked view, both formats
    np.testing.assert_array_equal(wide.grid.numpy(), st["grid"])
    for t in range(8):
        act = torch.from_numpy(util.random_actions(B, 3, seed=t))
        a, b = wide.step(act), comp.step(act)
        for x, y in zip(a, b):
            assert x.numpy().tobytes() == y.numpy().tobytes()
        np.testing.assert_array_equal(wide.grid.numpy(), comp.grid.numpy())
    sd = comp.state_dict()
    assert sd["spec"]["cell_bytes"] == 1 and sd["grid"].shape == (B, 10, 12, 3)
    other = BatchedMultiGridEnv(compact(spec), B, "cpu", backend=util.OracleBackend(compact(spec)))
    other.load_state_dict(sd)
    np.testing.assert_array_equal(other.cells.numpy(), comp.cells.numpy())


def test_what_compact_cells_do_not_serve_is_refused_by_name():
    sp = compact(EnvSpec(16, 16, 2, 7, max_steps=50))
    env = BatchedMultiGridEnv(sp, 4, "cpu", backend=util.OracleBackend(sp))
    st = util.random_state(sp, 4, seed=1)
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    act = torch.zeros((4, 2), dtype=torch.int8)
    out = env.rollout(act[None])                               # (round 6: the hook-free rollout / persistent launch run on compact cells)
    assert tuple(out["obs"].shape) == (1, 4, 2, 7, 7, 3)
    with pytest.raises(NotImplementedError, match="compact cells"):
        env.rollout(act[None], one_hot=True)
    oh, *_ = env.step(act, one_hot=True)                       # (round 6: the hook-free step writes one-hot observations on compact cells too)
    assert tuple(oh.shape) == (4, 2, 7, 7, 21)
    hooked = compact(EnvSpec(11, 6, 2, 7, max_steps=50, joint_reward=True, env_kind="blockedunlockpickup"))
    henv = BatchedMultiGridEnv(hooked, 4, "cpu", backend=util.OracleBackend(hooked))
    hst = util.random_state(hooked, 4, seed=2)
    henv.load_state(hst["grid"], hst["agents"], hst["rng"], hst["target"], hst["step_count"])
    with pytest.raises(NotImplementedError, match="compact cells"):
        henv.step(act, one_hot=True)                            # ... the hook envs (small grids) keep the 16-bit cells for it
    with pytest.raises(NotImplementedError, match="compact cells"):
        env.set_layout_generator("empty_fixed")
    with pytest.raises(NotImplementedError, match="compact cells"):
        henv.rollout(act[None])
    with pytest.raises(NotImplementedError, match="compact cells"):
        henv.persistent(4)
    with pytest.raises(ValueError):
        EnvSpec(8, 8, cell_bytes=4)
    bad = st["grid"].copy()
    bad[0, 3, 3] = (5, 1, 1)                                   # a key with a state: the compact format has no code for it
    with pytest.raises(ValueError, match="compact"):
        env.load_state(bad, st["agents"], st["rng"], st["target"], st["step_count"])


def test_c_abi_refuses_the_unsupported_entry_points_without_a_gpu():
    """Argument validation runs before any HIP call (tests/test_capi.py does the same for the other checks)."""
    import ctypes as C
    from multigrid_amd import _lib
    L = _lib.lib()
    sc = compact(EnvSpec(64, 64, 16, 9)).to_c()
    hsc = compact(EnvSpec(11, 6, 2, 7, env_kind="blockedunlockpickup")).to_c()
    assert L.mgx_rollout(C.byref(hsc), 8, 0, *([None] * 13)) == _lib.ERR_UNSUPPORTED         # (steps = 0: the spec check alone; hook envs)
    assert L.mgx_rollout(C.byref(sc), 8, 0, *([None] * 13)) != _lib.ERR_UNSUPPORTED          # (round 6: hook-free rollouts are served)
    key = (C.c_int32 * 16)()
    assert L.mgx_shape_key(C.byref(sc), 64, key) == _lib.ERR_UNSUPPORTED
    sc.cell_bytes = 5
    info = _lib.MgxLaunchInfo()
    assert L.mgx_launch_info(C.byref(sc), 64, C.byref(info)) == _lib.ERR_INVALID_ARGUMENT
    # two 64x64 envs per wavefront on compact cells, one on the 16-bit ones (the per-agent phases then run on 32 lanes, not 16)
    assert _lib.launch_info(compact(EnvSpec(64, 64, 16, 9)), 32768)["envs_per_wavefront"] == 2
    assert _lib.launch_info(EnvSpec(64, 64, 16, 9), 32768)["envs_per_wavefront"] == 1
    # (its shape instantiation: with the grids left to the caches at 128 MiB, streamed past them beyond that)
    assert _lib.launch_info(compact(EnvSpec(64, 64, 16, 9)), 32768)["fixed_shape"] == 6
    assert _lib.launch_info(compact(EnvSpec(64, 64, 16, 9)), 65536)["fixed_shape"] == 5


# ---------------------------------------------------------------------------------------------------- GPU: the kernels
gpu = pytest.mark.gpu

CASES = [
    ("c8_C5_64x64_a16_v9", EnvSpec(64, 64, 16, 9, max_steps=16384), 515, 6),
    ("c8_C2_empty16_a4_v7", EnvSpec(16, 16, 4, 7, max_steps=1024), 4096, 10),
    ("c8_bup_11x6_a2", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 3001, 10),
    ("c8_ragged_a3_v5_nooverlap", EnvSpec(9, 7, 3, 5, max_steps=50, allow_agent_overlap=False, failure_termination_mode="any"), 1001, 12),
    ("c8_a1_v3_seethrough", EnvSpec(8, 8, 1, 3, max_steps=30, see_through_walls=True), 777, 8),
    ("c8_a2_v9_seethrough", EnvSpec(10, 8, 2, 9, max_steps=30, see_through_walls=True), 130, 8),
    ("c8_a5_v11_all_joint", EnvSpec(13, 12, 5, 11, max_steps=40, success_termination_mode="all", joint_reward=True), 333, 8),
    ("c8_a7_v13", EnvSpec(20, 17, 7, 13, max_steps=40), 129, 5),
    ("c8_a2_v15", EnvSpec(24, 24, 2, 15, max_steps=40), 65, 5),
    ("c8_a32_v7", EnvSpec(12, 12, 32, 7, max_steps=40), 37, 5),
    ("c8_a17_v9_odd", EnvSpec(31, 23, 17, 9, max_steps=40), 203, 5),
    ("c8_single_env", EnvSpec(8, 8, 2, 7, max_steps=256), 1, 12),
    ("c8_max_grid_254x254_a2_v5", EnvSpec(254, 254, 2, 5, max_steps=20), 5, 3),
]


@gpu
@pytest.mark.parametrize("name,spec,B,T", CASES, ids=[c[0] for c in CASES])
def test_compact_random_states_vs_oracle(name, spec, B, T):
    from tests.test_hip_parity import test_random_states_vs_oracle
    test_random_states_vs_oracle(name, compact(spec), B, T)


THROUGHPUT = [
    ("c8t_a2_v9", EnvSpec(10, 8, 2, 9, max_steps=30), 40001, 4),
    ("c8t_a16_v9_48x48", EnvSpec(48, 48, 16, 9, max_steps=60), 6001, 3),
    ("c8t_bup_a2_v7", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 70003, 4),
    ("c8t_a5_v11", EnvSpec(13, 12, 5, 11, max_steps=40), 13001, 3),
    ("c8t_a7_v7", EnvSpec(12, 12, 7, 7, max_steps=40), 20011, 4),
]


@gpu
@pytest.mark.parametrize("name,spec,B,T", THROUGHPUT, ids=[c[0] for c in THROUGHPUT])
def test_compact_throughput_and_auto_reset_vs_oracle(name, spec, B, T):
    from tests.test_hip_parity import test_throughput_instantiations_vs_oracle
    test_throughput_instantiations_vs_oracle(name, compact(spec), B, T)


@gpu
def test_compact_equals_wide_on_the_same_states_and_actions():
    """Both formats through the HIP kernels, same state, same actions: every output and the whole state byte for byte."""
    spec = EnvSpec(40, 36, 9, 9, max_steps=200)
    B = 2003
    st = util.random_state(spec, B, seed=9)
    envs = []
    for sp in (spec, compact(spec)):
        env = BatchedMultiGridEnv(sp, B, DEV)
        env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
        envs.append(env)
    wide, comp = envs
    np.testing.assert_array_equal(comp.cells.cpu().numpy(), layouts.pack_cells8(st["grid"]))
    for t in range(12):
        act = torch.from_numpy(util.random_actions(B, spec.num_agents, seed=300 + t)).to(DEV)
        a, b = wide.step(act), comp.step(act)
        for x, y in zip(a, b):
            assert x.cpu().numpy().tobytes() == y.cpu().numpy().tobytes(), f"step {t}"
        np.testing.assert_array_equal(wide.grid.cpu().numpy(), comp.grid.cpu().numpy())
        np.testing.assert_array_equal(wide.agents.cpu().numpy(), comp.agents.cpu().numpy())
        # the compact cells stay canonical: what the packer makes of the unpacked grid (opaque bits up to date)
        np.testing.assert_array_equal(comp.cells.cpu().numpy(), layouts.pack_cells8(comp.grid.cpu().numpy()))
    np.testing.assert_array_equal(wide.full_obs().cpu().numpy(), comp.full_obs().cpu().numpy())
    wide.check_errors(); comp.check_errors()


@gpu
def test_compact_pack_unpack_check_kernels():
    import ctypes as C
    from multigrid_amd import _lib
    L = _lib.lib()
    spec = compact(EnvSpec(30, 20, 3, 7, max_steps=50))
    B = 257
    st = util.random_state(spec, B, seed=3)
    g3 = torch.from_numpy(st["grid"]).to(DEV)
    cells = torch.zeros((B, 20, 30), dtype=torch.uint8, device=DEV)
    bad = torch.zeros(2, dtype=torch.int32, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    assert L.mgx_pack_grid8_env(g3.data_ptr(), B, 20, 30, cells.data_ptr(), bad.data_ptr(), s) == 0
    assert bad.cpu().tolist() == [0, 0]
    np.testing.assert_array_equal(cells.cpu().numpy(), layouts.pack_cells8(st["grid"]))
    back = torch.zeros_like(g3)
    assert L.mgx_unpack_grid8(cells.data_ptr(), cells.numel(), back.data_ptr(), s) == 0
    np.testing.assert_array_equal(back.cpu().numpy(), st["grid"])
    # a state on a ball and a hole in the wall ring are counted
    g3[5, 4, 4] = torch.tensor([6, 1, 2], dtype=torch.uint8)
    g3[7, 0, 3] = torch.tensor([1, 0, 0], dtype=torch.uint8)
    assert L.mgx_pack_grid8_env(g3.data_ptr(), B, 20, 30, cells.data_ptr(), bad.data_ptr(), s) == 0
    assert bad.cpu().tolist() == [1, 1]
    # mgx_check_grid on compact state: clean, then a broken opaque bit and an agent on the ring
    sc = spec.to_c()
    chk = torch.tensor([0, 0, 0, 2 ** 31 - 1], dtype=torch.int32, device=DEV)
    good = torch.from_numpy(layouts.pack_cells8(st["grid"])).to(DEV)
    agents = torch.from_numpy(st["agents"]).to(DEV)
    assert L.mgx_check_grid(C.byref(sc), B, good.data_ptr(), agents.data_ptr(), chk.data_ptr(), s) == 0
    assert chk.cpu().tolist() == [0, 0, 0, 2 ** 31 - 1]
    good[11, 5, 5] = 0x82 ^ 0x80                               # a wall without its opaque bit
    agents[13, 0, 2] = 0
    assert L.mgx_check_grid(C.byref(sc), B, good.data_ptr(), agents.data_ptr(), chk.data_ptr(), s) == 0
    assert chk.cpu().tolist() == [1, 0, 1, 11]


@gpu
def test_compact_sub_shard_chains_equal_one_launch():
    spec = compact(EnvSpec(48, 48, 8, 9, max_steps=100))
    B = 4096
    st = util.random_state(spec, B, seed=21)
    a, b = (BatchedMultiGridEnv(spec, B, DEV) for _ in range(2))
    for env in (a, b):
        env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    for t in range(5):
        act = torch.from_numpy(util.random_actions(B, 8, seed=t)).to(DEV)
        a.step(act)
        b.step(act, sub_shards=2)
    b.join()
    for name in ("cells", "agents", "rng", "step_count", "obs", "reward", "terminated", "truncated"):
        assert getattr(a, name).cpu().numpy().tobytes() == getattr(b, name).cpu().numpy().tobytes(), name


@gpu
def test_c5_full_size_on_both_cell_formats_vs_oracle():
    """BASELINE.json configs[4] at its full size, every step, every output and the whole state, fused auto-reset: on the compact
    cells the bench times it on (two envs per wavefront, the shape-specialised kernel) and on the 16-bit cells."""
    from tests.test_full_size import run_vs_oracle
    wl = workloads.make("c5")
    assert wl.batch == 32768 and wl.spec.cell_bytes == 1
    env = wl.make_env(DEV)
    li = env.backend.launch_info(wl.batch)
    assert li["envs_per_wavefront"] == 2 and li["fixed_shape"] == 6, li
    run_vs_oracle(wl, T=8, seed=55)
    # ... and beyond 128 MiB of grids the streamed instantiation of the same shape (nt tile loads)
    wl_big = workloads.make("c5", global_batch=36864)
    assert wl_big.make_env(DEV).backend.launch_info(wl_big.batch)["fixed_shape"] == 5
    run_vs_oracle(wl_big, T=3, seed=56)
    wl2 = workloads.make("c5", cell_bytes=2)
    assert wl2.spec.cell_bytes == 2 and wl2.make_env(DEV).backend.launch_info(wl2.batch)["fixed_shape"] == 4
    run_vs_oracle(wl2, T=4, seed=56)


OH_CASES = [c for c in CASES if c[1].env_kind == "empty"]


@gpu
@pytest.mark.parametrize("name,spec,B,T", OH_CASES, ids=[c[0] + "_one_hot" for c in OH_CASES])
def test_compact_one_hot_step_vs_oracle(name, spec, B, T):
    """Round 6: the step with ONE-HOT output on compact cells (the decode table holds the cells' one-hot masks) == the oracle's step
    + OneHotObsWrapper.one_hot (multigrid/wrappers.py:158-190), every view size, and the state stays the compact one."""
    sp = compact(spec)
    st = util.random_state(sp, B, seed=zlib.crc32(name.encode()) % 10000, density=0.25)
    env = BatchedMultiGridEnv(sp, B, DEV)
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    ref = {k: v.copy() for k, v in st.items()}
    sd = spec.as_dict()
    o_ref, d_ref = ob.gen_obs_batch(sd, ref["grid"], ref["agents"], nthreads=8)
    oh, dirs = env.gen_obs(one_hot=True)                       # (two launches on compact cells: gen_obs, one_hot)
    np.testing.assert_array_equal(oh.cpu().numpy(), ob.one_hot(o_ref))
    for t in range(min(T, 6)):
        act = util.random_actions(B, spec.num_agents, seed=1000 + t)
        o_ref, d_ref, r_ref, te_ref, tr_ref = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act,
                                                            ref["target"], nthreads=8)
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


ROLL_CASES8 = [c for c in OH_CASES if c[1].width <= 64]


@gpu
@pytest.mark.parametrize("name,spec,B,T", ROLL_CASES8, ids=[c[0] + "_rollout_persistent" for c in ROLL_CASES8])
def test_compact_rollout_and_persistent_equal_repeated_steps(name, spec, B, T):
    """Round 6: mgx_rollout* and mgx_step_persistent on compact cells (the 4 KiB tile of a 64x64 env resident in LDS) == T x mgx_step on
    the same cells == the oracle (tests above), with the fused auto-reset."""
    sp = compact(spec)
    st = util.random_state(sp, B, seed=zlib.crc32(name.encode()) % 10000, density=0.25)
    pool = util.random_state(sp, 3, seed=5, terminated_p=0.0)
    T = min(T, 5)
    envs = []
    for _ in range(3):
        e = BatchedMultiGridEnv(sp, B, DEV, first_env=3)
        e.load_state(st["grid"], st["agents"], st["rng"], None, st["step_count"])
        e.set_layout_pool(pool["grid"], pool["agents"])
        e.step_count[::2] = sp.max_steps - 2
        envs.append(e)
    e_step, e_roll, e_per = envs
    acts = torch.from_numpy(np.stack([util.random_actions(B, spec.num_agents, seed=300 + t) for t in range(T)])).to(DEV)
    out = e_roll.rollout(acts, auto_reset=True)
    try:
        ps = e_per.persistent(max_steps=T, auto_reset=True)
        ps.__enter__()
    except Exception as exc:                                    # (batches beyond what is resident: the rollout alone)
        from multigrid_amd import _lib
        assert isinstance(exc, _lib.MgxError) and exc.code == _lib.ERR_UNSUPPORTED, exc
        ps = None
    for t in range(T):
        want = [x.clone() for x in e_step.step(acts[t], auto_reset=True)] + [e_step.was_reset.clone()]
        for k, n in enumerate(("obs", "dir", "reward", "terminated", "truncated", "was_reset")):
            assert torch.equal(out[n][t], want[k]), f"{name} rollout step {t}: {n}"
        if ps is not None:
            got = list(ps.step(acts[t])) + [e_per.was_reset]
            for k, (g, w) in enumerate(zip(got, want)):
                assert torch.equal(g, w), f"{name} persistent step {t}: output {k}"
    if ps is not None:
        ps.close()
        assert ps.timeouts == 0
    for e in (e_roll,) + ((e_per,) if ps is not None else ()):
        for n in ("cells", "agents", "rng", "step_count", "episode"):
            assert torch.equal(getattr(e, n), getattr(e_step, n)), n


@gpu
def test_c5_full_size_rollout_on_compact_cells_vs_oracle():
    """BASELINE.json configs[4] at its full size as ONE launch on compact cells, auto-reset fused in, against the oracle."""
    from tests.test_full_size import oracle_reset_done
    wl = workloads.make("c5")
    B, A, T = wl.batch, wl.spec.num_agents, 3
    env = wl.make_env(DEV, auto_reset=True)
    ref = dict(grid=wl.grid.copy(), agents=wl.agents.copy(), rng=wl.rng.copy(), step_count=np.zeros(B, np.int32), aux=None)
    episode = np.zeros(B, np.int32)
    sd = dataclasses.replace(wl.spec, cell_bytes=2).as_dict()
    acts = np.random.default_rng(8).integers(0, 7, size=(T, B, A)).astype(np.int8)
    out = env.rollout(torch.from_numpy(acts).to(DEV), auto_reset=True)
    for t in range(T):
        was = oracle_reset_done(wl, ref, episode)
        o, d, rw, te, tr = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], acts[t], None, nthreads=ob.max_threads())
        assert out["obs"][t].cpu().numpy().tobytes() == o.tobytes(), f"step {t}: obs"
        assert out["reward"][t].cpu().numpy().tobytes() == rw.tobytes(), f"step {t}: reward"
        np.testing.assert_array_equal(out["terminated"][t].cpu().numpy(), te)
        np.testing.assert_array_equal(out["was_reset"][t].cpu().numpy(), was)
    np.testing.assert_array_equal(env.grid.cpu().numpy(), ref["grid"])
    np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"])
    env.check_errors()


@gpu
def test_c5_full_size_one_hot_with_auto_reset_compact_equals_wide_and_oracle():
    """BASELINE.json configs[4] at its full size with one-hot output and the fused auto-reset -- the default RL path of the reference
    (rllib/__init__.py:110-111 wraps every env in OneHotObsWrapper): compact cells == 16-bit cells == the oracle's step + one_hot."""
    wl_c, wl_w = workloads.make("c5"), workloads.make("c5", cell_bytes=2)
    assert wl_c.spec.cell_bytes == 1 and wl_c.batch == 32768
    ec, ew = wl_c.make_env(DEV, auto_reset=True), wl_w.make_env(DEV, auto_reset=True)
    B, A = wl_c.batch, wl_c.spec.num_agents
    ref = dict(grid=wl_c.grid.copy(), agents=wl_c.agents.copy(), rng=wl_c.rng.copy(), step_count=np.zeros(B, np.int32), aux=None)
    sd = wl_w.spec.as_dict()
    r = np.random.default_rng(77)
    from tests.test_full_size import oracle_reset_done
    episode = np.zeros(B, np.int32)
    for t in range(4):
        act = r.integers(0, 7, size=(B, A)).astype(np.int8)
        oracle_reset_done(wl_w, ref, episode)                   # (the fused auto-reset, from its definition: include/mgx.h)
        o_ref, d_ref, r_ref, te_ref, tr_ref = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, None,
                                                            nthreads=ob.max_threads())
        a = torch.from_numpy(act).to(DEV)
        oc = ec.step(a, auto_reset=True, one_hot=True)
        ow = ew.step(a, auto_reset=True, one_hot=True)
        for n, x, y in zip(("one_hot", "dir", "reward", "terminated", "truncated"), oc, ow):
            assert torch.equal(x, y), f"step {t}: {n} (compact vs 16-bit cells)"
        assert oc[0].cpu().numpy().tobytes() == ob.one_hot(o_ref).tobytes(), f"step {t}: one-hot obs vs oracle"
        assert oc[2].cpu().numpy().tobytes() == r_ref.tobytes()
        np.testing.assert_array_equal(ec.grid.cpu().numpy(), ref["grid"])
        np.testing.assert_array_equal(ec.agents.cpu().numpy(), ref["agents"])
    ec.check_errors(); ew.check_errors()


# ---------------------------------------------------------------------------------------------------- byte grids (cell_bytes = 3)
def bytes_spec(spec: EnvSpec) -> EnvSpec:
    return dataclasses.replace(spec, cell_bytes=3)


BYTE_CASES = [
    ("b3_C2_empty16_a4_v7", EnvSpec(16, 16, 4, 7, max_steps=1024), 4096, 10),
    ("b3_bup_11x6_a2", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 3001, 10),     # 198 bytes per env: unaligned
    ("b3_ragged_a3_v5_nooverlap", EnvSpec(9, 7, 3, 5, max_steps=50, allow_agent_overlap=False, failure_termination_mode="any"), 1001, 10),
    ("b3_a1_v3_seethrough", EnvSpec(8, 8, 1, 3, max_steps=30, see_through_walls=True), 777, 6),
    ("b3_C5_64x64_a16_v9", EnvSpec(64, 64, 16, 9, max_steps=16384), 300, 5),
    ("b3_a5_v11_all_joint", EnvSpec(13, 12, 5, 11, max_steps=40, success_termination_mode="all", joint_reward=True), 333, 6),
    ("b3_a2_v15", EnvSpec(24, 24, 2, 15, max_steps=40), 65, 4),
    ("b3_single_env", EnvSpec(8, 8, 2, 7, max_steps=256), 1, 8),
    ("b3_big_grid_200x180_a3_v7", EnvSpec(200, 180, 3, 7, max_steps=40), 9, 3),
]


@gpu
@pytest.mark.parametrize("name,spec,B,T", BYTE_CASES, ids=[c[0] for c in BYTE_CASES])
def test_byte_grid_random_states_vs_oracle(name, spec, B, T):
    """MgxSpec.cell_bytes = 3: the grid tensor IS the reference's (type, color, state) bytes u8[B,H,W,3]; the step kernel packs it
    into its LDS tile as it loads it and writes changed cells back as three bytes."""
    from tests.test_hip_parity import test_random_states_vs_oracle
    test_random_states_vs_oracle(name, bytes_spec(spec), B, T)


@gpu
@pytest.mark.parametrize("name,spec,B,T", [("b3t_a3_v5", EnvSpec(9, 7, 3, 5, max_steps=50), 50001, 4),
                                           ("b3t_bup_a2_v7", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 70003, 4),
                                           ("b3t_a2_v9", EnvSpec(10, 8, 2, 9, max_steps=30), 40001, 3)], ids=["a3_v5", "bup", "a2_v9"])
def test_byte_grid_throughput_and_auto_reset_vs_oracle(name, spec, B, T):
    from tests.test_hip_parity import test_throughput_instantiations_vs_oracle
    test_throughput_instantiations_vs_oracle(name, bytes_spec(spec), B, T)


@gpu
def test_byte_grid_c4_full_size_and_filled_boxes_and_the_ring_count():
    from tests.test_full_size import run_vs_oracle
    wl = workloads.make("c4", cell_bytes=3)
    assert wl.spec.cell_bytes == 3
    run_vs_oracle(wl, T=10, seed=41)
    # boxes that hold things travel in the state byte's upper bits
    spec = bytes_spec(EnvSpec(12, 12, 4, 7, max_steps=200))
    B = 3000
    st = util.random_state(spec, B, seed=8, density=0.35, carry_p=0.5, box_contents_p=0.7)
    env = BatchedMultiGridEnv(spec, B, DEV)
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    ref = {k: v.copy() for k, v in st.items()}
    r = np.random.default_rng(5)
    for t in range(10):
        act = r.choice(7, size=(B, 4), p=[0.1, 0.1, 0.25, 0.2, 0.15, 0.18, 0.02]).astype(np.int8)
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["target"], nthreads=8)
        got = env.step(torch.from_numpy(act).to(DEV))
        for g, w in zip(got, want):
            assert g.cpu().numpy().tobytes() == w.tobytes(), f"step {t}"
        np.testing.assert_array_equal(env.cells.cpu().numpy(), ref["grid"], err_msg=f"step {t}")
    env.check_errors()
    # a hole in the wall ring and a value the packed format cannot hold are counted by the kernel (MgxStepArgs.grid_bad)
    env.cells[5, 0, 3] = torch.tensor([1, 0, 0], dtype=torch.uint8, device=DEV)
    env.cells[9, 4, 4] = torch.tensor([5, 9, 0], dtype=torch.uint8, device=DEV)
    env.step(torch.zeros((B, 4), dtype=torch.int8, device=DEV))
    with pytest.raises(ValueError, match="1 outer-ring cell.*1 cell value"):
        env.check_errors()
    # every edge of the ring is looked at, a corner once; grids whose ring is longer than a wavefront (2W + 2(H-2) > 64) too
    for W_, H_, B_ in ((12, 12, 300), (30, 21, 70), (7, 5, 1000)):
        sp = bytes_spec(EnvSpec(W_, H_, 2, 5, max_steps=50))
        st2 = util.random_state(sp, B_, seed=3, density=0.2)
        e2 = BatchedMultiGridEnv(sp, B_, DEV)
        e2.load_state(st2["grid"], st2["agents"], st2["rng"], st2["target"], st2["step_count"])
        e2.step(torch.zeros((B_, 2), dtype=torch.int8, device=DEV)); e2.check_errors()
        hole = torch.tensor([1, 0, 0], dtype=torch.uint8, device=DEV)
        holes = [(0, 0, 0), (1, 0, W_ - 1), (2, H_ - 1, 0), (3, H_ - 1, W_ - 1), (4, 0, W_ // 2), (5, H_ - 1, W_ // 2), (6, H_ // 2, 0),
                 (7, H_ // 2, W_ - 1), (B_ - 1, 1, 0), (B_ - 1, H_ - 2, W_ - 1), (B_ - 1, 0, 1)]
        for b_, y_, x_ in holes:
            e2.cells[b_, y_, x_] = hole
        e2.cells[8, H_ // 2, W_ // 2] = hole                         # (an interior cell: not the ring's business)
        e2.step(torch.zeros((B_, 2), dtype=torch.int8, device=DEV))
        with pytest.raises(ValueError, match=f"{len(holes)} outer-ring cell"):
            e2.check_errors()
