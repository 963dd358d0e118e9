"""GPU parity: the fused HIP kernel vs (a) the golden vectors recorded from the real reference and (b) the CPU
oracle on seeded random states.  Bit-exact on every output and on the post-step state.  Calls go through the
C ABI (libmgx.so) via BatchedMultiGridEnv / torch.ops.mgx."""
import zlib

import numpy as np
import pytest
import torch

from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts
from oracle import binding as ob
from tests import util

pytestmark = pytest.mark.gpu


def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return "cuda:0"


@pytest.mark.parametrize("path", util.GOLDEN, ids=util.GOLDEN_IDS)
@pytest.mark.parametrize("replicas", [1, 5])
def test_golden_replay(path, replicas):
    z, d, spec = util.load_golden(path)
    env = BatchedMultiGridEnv(spec, replicas, dev())
    env.load_state(layouts.grid_to_product(z["grid0"]), layouts.pack_agents(z["agents0"]),
                   rng=util.rng_words_lohi(z["rng0"]), aux=util.golden_aux(d))
    obs, dirs = env.gen_obs()
    for b in range(replicas):
        np.testing.assert_array_equal(obs[b].cpu().numpy(), z["obs0"])
        np.testing.assert_array_equal(dirs[b].cpu().numpy(), z["dir0"])
    T = z["actions"].shape[0]
    for t in range(T):
        act = torch.from_numpy(np.repeat(z["actions"][t][None], replicas, 0)).to(dev())
        ho = None
        if "hook_order" in z.files:                          # the dict order the reference was stepped with
            ho = torch.from_numpy(np.repeat(z["hook_order"][t][None], replicas, 0)).to(dev())
        obs, dirs, rew, term, trunc = env.step(act, hook_order=ho)
        ctx = f"step {t}"
        for b in (0, replicas - 1):
            np.testing.assert_array_equal(obs[b].cpu().numpy(), z["obs"][t], err_msg=ctx)
            np.testing.assert_array_equal(dirs[b].cpu().numpy(), z["direction"][t], err_msg=ctx)
            assert rew[b].cpu().numpy().tobytes() == z["reward"][t].tobytes(), ctx
            np.testing.assert_array_equal(term[b].cpu().numpy(), z["terminated"][t], err_msg=ctx)
            assert int(trunc[b]) == int(z["truncated"][t]), ctx
            np.testing.assert_array_equal(layouts.grid_from_product(env.grid[b].cpu().numpy()),
                                          z["grid"][t].astype(np.int64), err_msg=ctx)
            np.testing.assert_array_equal(layouts.unpack_agents(env.agents[b].cpu().numpy()),
                                          z["agents"][t].astype(np.int64), err_msg=ctx)
    env.check_errors()
    np.testing.assert_array_equal(env.rng[0].cpu().numpy().view(np.uint64), util.rng_words_lohi(z["rng_final"]))
    assert int(env.step_count[0]) == T


CASES = [
    # (name, spec, B, T)
    ("C2_empty16_a4_v7", EnvSpec(16, 16, 4, 7, max_steps=1024), 4096, 24),
    ("C3_bup_11x6_a2", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 16384, 12),
    ("C5_64x64_a16_v9", EnvSpec(64, 64, 16, 9, max_steps=16384), 512, 6),
    ("ragged_a3_v5_nooverlap", EnvSpec(9, 7, 3, 5, max_steps=50, allow_agent_overlap=False,
                                       failure_termination_mode="any"), 1001, 16),
    ("a1_v3_seethrough", EnvSpec(8, 8, 1, 3, max_steps=30, see_through_walls=True), 777, 12),
    ("a3_v7_seethrough", EnvSpec(8, 9, 3, 7, max_steps=30, see_through_walls=True), 515, 10),
    ("a2_v9_seethrough", EnvSpec(10, 8, 2, 9, max_steps=30, see_through_walls=True), 130, 8),
    ("a5_v11_all_joint", EnvSpec(13, 12, 5, 11, max_steps=40, success_termination_mode="all", joint_reward=True), 333, 10),
    ("a7_v13", EnvSpec(20, 17, 7, 13, max_steps=40), 129, 6),
    ("a2_v15", EnvSpec(24, 24, 2, 15, max_steps=40), 65, 6),
    ("a32_v7", EnvSpec(12, 12, 32, 7, max_steps=40), 37, 6),
    ("single_env", EnvSpec(8, 8, 2, 7, max_steps=256), 1, 20),
    # the largest grids the format allows (sides <= 254: DESIGN.md section 7): one env's tile is 72 / 129 KB of a CU's 160 KB of LDS
    ("big_grid_200x180_a3_v7", EnvSpec(200, 180, 3, 7, max_steps=40), 9, 4),
    ("max_grid_254x254_a2_v5", EnvSpec(254, 254, 2, 5, max_steps=20), 5, 3),
]


@pytest.mark.parametrize("name,spec,B,T", CASES, ids=[c[0] for c in CASES])
def test_random_states_vs_oracle(name, spec, B, T):
    st = util.random_state(spec, B, seed=zlib.crc32(name.encode()) % 10000)
    env = BatchedMultiGridEnv(spec, B, dev())
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    ref = {k: v.copy() for k, v in st.items()}
    sd = spec.as_dict()
    o_ref, d_ref = ob.gen_obs_batch(sd, ref["grid"], ref["agents"], nthreads=8)
    obs, dirs = env.gen_obs()
    np.testing.assert_array_equal(obs.cpu().numpy(), o_ref)
    np.testing.assert_array_equal(dirs.cpu().numpy(), d_ref)
    for t in range(T):
        act = util.random_actions(B, spec.num_agents, seed=1000 + t)
        o_ref, d_ref, r_ref, te_ref, tr_ref = ob.step_batch(
            sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["target"], nthreads=8)
        obs, dirs, rew, term, trunc = env.step(torch.from_numpy(act).to(dev()))
        ctx = f"{name} step {t}"
        np.testing.assert_array_equal(env.grid.cpu().numpy(), ref["grid"], err_msg=ctx)
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"], err_msg=ctx)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_ref, err_msg=ctx)
        np.testing.assert_array_equal(dirs.cpu().numpy(), d_ref, err_msg=ctx)
        assert rew.cpu().numpy().tobytes() == r_ref.tobytes(), ctx
        np.testing.assert_array_equal(term.cpu().numpy(), te_ref, err_msg=ctx)
        np.testing.assert_array_equal(trunc.cpu().numpy(), tr_ref, err_msg=ctx)
        np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref["step_count"], err_msg=ctx)
        if spec.num_agents > 1:
            np.testing.assert_array_equal(env.rng.cpu().numpy().view(np.uint64), ref["rng"], err_msg=ctx)
    env.check_errors()


# The library has two families of instantiations (multigrid_amd/csrc/mgx_fused.h): launches of <= 2048 wavefronts take the
# latency ones (LDS-DMA tile loads, 32 view slots, one cell per register), larger launches the throughput ones (64 view slots
# for views up to 7x7, two slots' packed cells per register).  The cases above are small; these are sized to land in the
# throughput instantiations with ragged batches, odd agent counts (partly filled slot groups), every view width class, a hook
# env and the fused auto-reset.
THROUGHPUT_CASES = [
    ("t_a3_v5_nooverlap", EnvSpec(9, 7, 3, 5, max_steps=50, allow_agent_overlap=False, failure_termination_mode="any"), 50001, 5),
    ("t_a1_v3_seethrough", EnvSpec(8, 8, 1, 3, max_steps=30, see_through_walls=True), 140001, 4),
    ("t_bup_a2_v7", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 70003, 5),
    ("t_a7_v7", EnvSpec(12, 12, 7, 7, max_steps=40), 20011, 5),
    ("t_a5_v7_all", EnvSpec(16, 16, 5, 7, max_steps=40, success_termination_mode="all", joint_reward=True), 26003, 5),
    ("t_a2_v9", EnvSpec(10, 8, 2, 9, max_steps=30), 40001, 4),
    ("t_a5_v11", EnvSpec(13, 12, 5, 11, max_steps=40), 13001, 4),
    ("t_a2_v15", EnvSpec(24, 24, 2, 15, max_steps=40), 40003, 3),
]


@pytest.mark.parametrize("name,spec,B,T", THROUGHPUT_CASES, ids=[c[0] for c in THROUGHPUT_CASES])
def test_throughput_instantiations_vs_oracle(name, spec, B, T):
    li = BatchedMultiGridEnv(spec, 1, dev()).backend.launch_info(B)
    assert -(-B // li["envs_per_wavefront"]) > 2048, li                  # (else the latency instantiation would run)
    test_random_states_vs_oracle(name, spec, B, T)
    # ... and through the fused auto-reset (restarts emulated in numpy from the definition, include/mgx.h)
    st = util.random_state(spec, B, seed=77, terminated_p=0.3)
    st["step_count"][::3] = spec.max_steps - 1
    K = 5
    pool = util.random_state(spec, K, seed=78, terminated_p=0.0, density=0.3)
    env = BatchedMultiGridEnv(spec, B, dev(), first_env=123)
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    env.set_layout_pool(pool["grid"], pool["agents"], pool["target"] if spec.env_kind != "empty" else None)
    ref = {k: v.copy() for k, v in st.items()}
    episode = np.zeros(B, np.int64)
    for t in range(3):
        done = (ref["agents"][:, :, 4] != 0).all(axis=1) | (ref["step_count"] >= spec.max_steps)
        lay = (123 + np.arange(B) + episode * 7919) % K
        ref["grid"][done] = pool["grid"][lay[done]]; ref["agents"][done] = pool["agents"][lay[done]]
        if spec.env_kind != "empty":
            ref["target"][done] = pool["target"][lay[done]]
        ref["step_count"][done] = 0; episode[done] += 1
        act = util.random_actions(B, spec.num_agents, seed=2000 + t)
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["target"], nthreads=8)
        got = env.step(torch.from_numpy(act).to(dev()), auto_reset=True)
        ctx = f"{name} auto-reset step {t}"
        np.testing.assert_array_equal(env.was_reset.cpu().numpy(), done.astype(np.uint8), err_msg=ctx)
        for g, w in zip(got, want):
            assert g.cpu().numpy().tobytes() == w.tobytes(), ctx
        np.testing.assert_array_equal(env.grid.cpu().numpy(), ref["grid"], err_msg=ctx)
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"], err_msg=ctx)
    assert int(episode.sum()) > B // 4
    env.check_errors()


BOX_CASES = [
    ("boxes_a4_v7", EnvSpec(12, 12, 4, 7, max_steps=200), 3000, 16),
    ("boxes_a2_v9_nooverlap", EnvSpec(14, 9, 2, 9, max_steps=200, allow_agent_overlap=False), 777, 12),
    ("boxes_a3_v5_throughput", EnvSpec(8, 8, 3, 5, max_steps=200), 60001, 6),
    ("boxes_a16_v9", EnvSpec(40, 40, 16, 9, max_steps=200), 300, 6),
]


@pytest.mark.parametrize("name,spec,B,T", BOX_CASES, ids=[c[0] for c in BOX_CASES])
def test_boxes_that_hold_things_vs_oracle(name, spec, B, T):
    """Box.contains (multigrid/core/world_object.py:574-605; include/mgx.h "BOX CONTENTS"): random states dense in boxes, most of
    them -- on the grid and in the agents' hands -- holding something; pickup-, drop- and toggle-heavy actions.  Every output and
    the whole state (contents included) against the oracle, whose restatement is pinned by the reference's `boxkey_a3` fixture."""
    st = util.random_state(spec, B, seed=zlib.crc32(name.encode()) % 10000, density=0.35, carry_p=0.5, box_contents_p=0.7)
    assert (st["grid"][..., 2] > 3).sum() > B // 4 and (st["agents"][..., 7] > 3).sum() > B // 50
    env = BatchedMultiGridEnv(spec, B, dev())
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    ref = {k: v.copy() for k, v in st.items()}
    r = np.random.default_rng(5)
    n_opened = 0
    for t in range(T):
        act = r.choice(7, size=(B, spec.num_agents), p=[0.1, 0.1, 0.25, 0.2, 0.15, 0.18, 0.02]).astype(np.int8)
        boxes_before = int((ref["grid"][..., 0] == 7).sum() + (ref["agents"][..., 5] == 7).sum())
        want = ob.step_batch(spec.as_dict(), ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["target"], nthreads=8)
        n_opened += boxes_before - int((ref["grid"][..., 0] == 7).sum() + (ref["agents"][..., 5] == 7).sum())
        got = env.step(torch.from_numpy(act).to(dev()))
        for g, w in zip(got, want):
            assert g.cpu().numpy().tobytes() == w.tobytes(), f"{name} step {t}"
        np.testing.assert_array_equal(env.grid.cpu().numpy(), ref["grid"], err_msg=f"{name} step {t}")
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"], err_msg=f"{name} step {t}")
    assert n_opened > B // 20, n_opened
    assert (env.obs[..., 2] > 3).sum() == 0                        # no observation ever shows a content
    env.check_errors()


def test_torch_ops_registered_and_match():
    import multigrid_amd.ops as ops
    spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
    st = util.random_state(spec, 64, seed=5)
    g = util.dev_cells(st["grid"], dev()); a = torch.from_numpy(st["agents"]).to(dev())
    obs, dirs = torch.ops.mgx.gen_obs(g, a, ops.spec_to_ints(spec))
    o_ref, d_ref = ob.gen_obs_batch(spec.as_dict(), st["grid"], st["agents"])
    np.testing.assert_array_equal(obs.cpu().numpy(), o_ref)
    np.testing.assert_array_equal(dirs.cpu().numpy(), d_ref)
    with pytest.raises(NotImplementedError):
        torch.ops.mgx.gen_obs(g.cpu(), a.cpu(), ops.spec_to_ints(spec))


def test_unknown_action_is_reported():
    spec = EnvSpec(8, 8, 2, 7, max_steps=256)
    st = util.random_state(spec, 32, seed=9, terminated_p=0.0)
    env = BatchedMultiGridEnv(spec, 32, dev())
    env.load_state(st["grid"], st["agents"], st["rng"])
    act = torch.zeros((32, 2), dtype=torch.int8, device=dev())
    act[17, 1] = 9
    env.step(act)
    with pytest.raises(ValueError, match="Unknown action"):
        env.check_errors()
    env.step(torch.zeros((32, 2), dtype=torch.int8, device=dev()))
    env.check_errors()


@pytest.mark.parametrize("path", util.LAYOUT_GOLDEN, ids=util.LAYOUT_IDS)
def test_dict_api_reset_sequence_on_gpu(path):
    """multigrid_amd.make(...) on the HIP backend reproduces the reference's reset()/step() sequence."""
    import multigrid_amd as mg
    z = np.load(path)
    name = util.LAYOUT_ENV_IDS[path.split("/")[-1][:-4]]
    A = z["agents0"].shape[1]
    env = mg.make(name, agents=A, layout_seed=int(z["construct_seed"]), device=dev())
    for k, sd in enumerate(z["reset_seeds"]):
        obs, _ = env.reset(seed=None if sd < 0 else int(sd))
        np.testing.assert_array_equal(env.grid.state, z["grid0"][k].astype(np.int64))
        np.testing.assert_array_equal(env.agent_states, z["agents0"][k].astype(np.int64))
        np.testing.assert_array_equal(env._benv.rng[0].cpu().numpy().view(np.uint64), util.rng_words_lohi(z["rng0"][k]))
        for i in range(A):
            np.testing.assert_array_equal(obs[i]["image"], z["obs0"][k][i])
        for t in range(5):
            env.step({i: int(z["actions"][k][t, i]) for i in range(A)})


def test_dict_api_replays_c1_on_gpu():
    """BASELINE.json configs[0]: MultiGrid-Empty-8x8-v0, agents=2, batch=1 -- from reset(seed) alone."""
    import multigrid_amd as mg
    z, d, spec = util.load_golden([p for p in util.GOLDEN if "empty8_a2_seed0" in p][0])
    env = mg.RLlibWrapper(mg.make("MultiGrid-Empty-8x8-v0", agents=2, device=dev()))
    obs, _ = env.reset(seed=d["seed"])
    for t in range(z["actions"].shape[0]):
        obs, rew, term, trunc, _ = env.step({i: int(a) for i, a in enumerate(z["actions"][t])})
        for i in range(2):
            np.testing.assert_array_equal(obs[i]["image"], z["obs"][t][i])
            assert obs[i]["direction"] == z["direction"][t][i] and rew[i] == z["reward"][t][i]
            assert term[i] == bool(z["terminated"][t][i]) and trunc[i] == bool(z["truncated"][t])
        assert term["__all__"] == all(bool(x) for x in z["terminated"][t])
    with pytest.raises(ValueError, match="Unknown action"):
        env.step({0: 9})


def test_full_size_properties_c4_shape():
    """BASELINE.json full size (Empty-16x16, 4 agents, 65536 envs): size-independent properties instead of an oracle
    run: (1) determinism / shard invariance: two halves stepped separately == the whole batch; (2) replicated envs
    with identical RNG and actions stay identical; (3) gen_obs is idempotent and equals the step's obs."""
    spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
    B = 65536
    grid, agents = layouts.empty_layout(16, 4)
    whole = BatchedMultiGridEnv(spec, B, dev()); whole.load_state(grid, agents); whole.seed_synthetic(7)
    lo = BatchedMultiGridEnv(spec, B // 2, dev(), first_env=0); lo.load_state(grid, agents); lo.seed_synthetic(7)
    hi = BatchedMultiGridEnv(spec, B // 2, dev(), first_env=B // 2); hi.load_state(grid, agents); hi.seed_synthetic(7)
    g = torch.Generator(device=dev()); g.manual_seed(5)
    for t in range(16):
        act = torch.randint(0, 7, (B, 4), dtype=torch.int8, device=dev(), generator=g)
        ow = whole.step(act)
        ol = lo.step(act[:B // 2].contiguous()); oh = hi.step(act[B // 2:].contiguous())
        for w, a, b in zip(ow, ol, oh):
            assert torch.equal(w, torch.cat([a, b]))
    assert torch.equal(whole.grid, torch.cat([lo.grid, hi.grid]))
    obs_step = whole.obs.clone()
    o1, _ = whole.gen_obs(); o1 = o1.clone()
    o2, _ = whole.gen_obs()
    assert torch.equal(o1, o2) and torch.equal(o1, obs_step)
    # replicas: same RNG words + same actions -> same trajectories
    rep = BatchedMultiGridEnv(spec, 4096, dev()); rep.load_state(grid, agents)
    rep.rng.copy_(whole.rng[:1].expand(4096, 4))
    for t in range(32):
        a1 = torch.randint(0, 7, (1, 4), dtype=torch.int8, device=dev(), generator=g).expand(4096, 4).contiguous()
        o = rep.step(a1)
        for x in o:
            assert bool((x == x[:1]).all())
    whole.check_errors(); rep.check_errors()


def test_tensors_beyond_4gib_address_correctly():
    """8.5 M envs of the C2 shape: obs is 5.0 GB and the packed grid 4.35 GB, so byte offsets leave 32 bits.  The first and the last
    2048 envs (with an odd, ragged tail) must come out exactly as when those envs are stepped on their own -- env-local
    results cannot depend on where the env sits in the batch."""
    spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
    B, n = 8_500_003, 2048
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * (1 << 30):
        pytest.skip("needs ~14 GB of free HBM")
    grid, agents = layouts.empty_layout(16, 4)
    big = BatchedMultiGridEnv(spec, B, dev()); big.load_state(grid, agents); big.seed_synthetic(11)
    head = BatchedMultiGridEnv(spec, n, dev(), first_env=0); head.load_state(grid, agents); head.seed_synthetic(11)
    tail = BatchedMultiGridEnv(spec, n, dev(), first_env=B - n); tail.load_state(grid, agents); tail.seed_synthetic(11)
    assert big.obs.numel() > (1 << 32) and big.cells.numel() * 2 > (1 << 32)
    g = torch.Generator(device=dev()); g.manual_seed(3)
    for t in range(6):
        act = torch.randint(0, 7, (B, 4), dtype=torch.int8, device=dev(), generator=g)
        ob_ = big.step(act)
        oh = head.step(act[:n].contiguous()); ot = tail.step(act[B - n:].contiguous())
        for w, a, b in zip(ob_, oh, ot):
            assert torch.equal(w[:n], a) and torch.equal(w[B - n:], b)
    assert torch.equal(big.cells[B - n:], tail.cells) and torch.equal(big.agents[B - n:], tail.agents)
    assert torch.equal(big.rng[B - n:], tail.rng) and torch.equal(big.step_count[B - n:], tail.step_count)
    o1, _ = big.gen_obs()
    assert torch.equal(o1[B - n:], tail.obs) and torch.equal(o1[:n], head.obs)
    big.check_errors()
    del big, head, tail
    torch.cuda.empty_cache()


@pytest.mark.parametrize("path", util.WRAPPER_GOLDEN, ids=util.WRAPPER_IDS)
def test_wrapper_kernels_vs_reference_goldens(path):
    import json
    z = np.load(path)
    spec = EnvSpec.from_dict(json.loads(str(z["spec_json"])))
    T = z["obs"].shape[0]
    env = BatchedMultiGridEnv(spec, T, dev())
    env.load_state(layouts.grid_to_product(z["grid"]), layouts.pack_agents(z["agents"]), validate=False)
    obs, _ = env.gen_obs()
    np.testing.assert_array_equal(obs.cpu().numpy(), z["obs"])
    np.testing.assert_array_equal(env.one_hot_obs().cpu().numpy(), z["one_hot"])
    np.testing.assert_array_equal(env.full_obs().cpu().numpy(), z["full"])


def test_one_hot_and_full_obs_vs_oracle_at_scale():
    import multigrid_amd.ops as ops
    spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
    B = 3001
    st = util.random_state(spec, B, seed=77, terminated_p=0.2)
    env = BatchedMultiGridEnv(spec, B, dev())
    env.load_state(st["grid"], st["agents"], st["rng"])
    obs, _ = env.gen_obs()
    want = ob.one_hot(obs.cpu().numpy())
    np.testing.assert_array_equal(env.one_hot_obs().cpu().numpy(), want)
    np.testing.assert_array_equal(torch.ops.mgx.one_hot(obs, [11, 6, 4]).cpu().numpy(), want)
    # ragged sizes / other channel counts through the raw op
    for n, dims in ((1, (11, 6, 4)), (5, (11, 6, 4)), (1000003, (11, 6, 4)), (777, (16, 8, 8)), (333, (3, 2, 2))):
        r = np.random.default_rng(n)
        cells = np.stack([r.integers(0, d, size=n) for d in dims], axis=1).astype(np.uint8)
        got = torch.ops.mgx.one_hot(torch.from_numpy(cells).to(dev()), list(dims)).cpu().numpy()
        np.testing.assert_array_equal(got, ob.one_hot(cells, dims))
    full = env.full_obs().cpu().numpy()
    g, a = st["grid"], st["agents"]
    for b in range(0, B, 97):
        np.testing.assert_array_equal(full[b], ob.full_obs(layouts.grid_from_product(g[b]), layouts.unpack_agents(a[b])))
    got = torch.ops.mgx.full_obs(env.cells, env.agents, ops.spec_to_ints(spec))
    assert torch.equal(got, env.full_obs())


AUX_SHAPES = [(5, 5, 3, 7), (9, 7, 2, 1000), (11, 6, 2, 4097), (16, 16, 4, 513), (64, 64, 16, 37), (100, 90, 5, 9)]


@pytest.mark.parametrize("W,H,A,B", AUX_SHAPES, ids=[f"{w}x{h}_a{a}_b{b}" for w, h, a, b in AUX_SHAPES])
def test_full_obs_and_reset_done_odd_shapes(W, H, A, B):
    """Layout sizes that are not multiples of 16 / 4 / 2 bytes, ragged batches, tiles larger than one LDS pass."""
    spec = EnvSpec(W, H, A, 7, max_steps=9)
    st = util.random_state(spec, B, seed=W * 1000 + H, terminated_p=0.3)
    env = BatchedMultiGridEnv(spec, B, dev(), first_env=5)
    env.load_state(st["grid"], st["agents"], st["rng"])
    full = env.full_obs().cpu().numpy()
    for b in sorted(set(list(range(0, B, max(1, B // 23))) + [B - 1])):
        np.testing.assert_array_equal(full[b], ob.full_obs(layouts.grid_from_product(st["grid"][b]),
                                                           layouts.unpack_agents(st["agents"][b])))
    # auto-reset: a random subset is over (truncated or all agents terminated), the rest must stay untouched
    K = 7
    pool = util.random_state(spec, K, seed=99, terminated_p=0.0)
    env.set_layout_pool(pool["grid"], pool["agents"])
    r = np.random.default_rng(B)
    sc = np.where(r.random(B) < 0.4, spec.max_steps, r.integers(0, spec.max_steps, size=B)).astype(np.int32)
    env.step_count.copy_(torch.from_numpy(sc))
    all_term = st["agents"][:, :, 4].min(axis=1) > 0
    done = all_term | (sc >= spec.max_steps)
    was = env.reset_done().cpu().numpy().astype(bool)
    np.testing.assert_array_equal(was, done)
    k = (5 + np.arange(B)) % K
    want_grid = np.where(done[:, None, None, None], pool["grid"][k], st["grid"])
    want_agents = np.where(done[:, None, None], pool["agents"][k], st["agents"])
    np.testing.assert_array_equal(env.grid.cpu().numpy(), want_grid)
    np.testing.assert_array_equal(env.agents.cpu().numpy(), want_agents)
    np.testing.assert_array_equal(env.step_count.cpu().numpy(), np.where(done, 0, sc))
    np.testing.assert_array_equal(env.episode.cpu().numpy(), done.astype(np.int32))


def test_reset_done_on_gpu_matches_definition():
    spec = EnvSpec(11, 6, 2, 7, max_steps=6, joint_reward=True, env_kind="blockedunlockpickup")
    B, K, first = 1000, 17, 12345
    r = np.random.default_rng(3)
    pool = [layouts.blockedunlockpickup_layout(6, 2, r, r) for _ in range(K)]
    pg, pa, pt = (np.stack([p[i] for p in pool]) for i in range(3))
    env = BatchedMultiGridEnv(spec, B, dev(), first_env=first)
    pt = np.stack([layouts.make_aux("blockedunlockpickup", pg[k], pt[k]) for k in range(K)])
    env.load_state(pg[0], pa[0], aux=pt[0]); env.seed_synthetic(2)
    env.set_layout_pool(pg, pa, pt)
    g = torch.Generator(device=dev()); g.manual_seed(1)
    for ep in range(3):
        for t in range(6):
            env.step(torch.randint(0, 7, (B, 2), dtype=torch.int8, device=dev(), generator=g))
            if t < 5:
                assert int(env.reset_done().sum()) == int((env.terminated.min(dim=1).values > 0).sum())
        was = env.reset_done().cpu().numpy()
        assert was.all()
        k = (first + np.arange(B) + np.maximum(env.episode.cpu().numpy() - 1, 0) * 7919) % K
        np.testing.assert_array_equal(env.grid.cpu().numpy(), pg[k])
        np.testing.assert_array_equal(env.agents.cpu().numpy(), pa[k])
        np.testing.assert_array_equal(env.aux.cpu().numpy(), pt[k])
        assert int(env.step_count.sum()) == 0
    env.check_errors()


AR_CASES = [
    ("empty8_a2", EnvSpec(8, 8, 2, 7, max_steps=7), 777, 40),
    ("empty16_a4_ragged", EnvSpec(16, 16, 4, 7, max_steps=5), 4099, 24),
    ("odd_9x7_a3_v5", EnvSpec(9, 7, 3, 5, max_steps=6, failure_termination_mode="any"), 515, 30),
    ("bup_a2", EnvSpec(11, 6, 2, 7, max_steps=9, joint_reward=True, env_kind="blockedunlockpickup"), 1500, 40),
    ("a16_v9_64x64", EnvSpec(64, 64, 16, 9, max_steps=4), 40, 10),
]


def _ar_env(spec, B, seed):
    """Env + layout pool: random walled states (objects, lava, goals -> early terminations) for the plain kinds, real
    BlockedUnlockPickup layouts for the hook kind."""
    K = 5
    if spec.env_kind == "blockedunlockpickup":
        r = np.random.default_rng(seed)
        pool = [layouts.blockedunlockpickup_layout(6, 2, r, r) for _ in range(K)]
        pg, pa = np.stack([p[0] for p in pool]), np.stack([p[1] for p in pool])
        pt = np.stack([layouts.make_aux("blockedunlockpickup", p[0], p[2]) for p in pool])
    else:
        st = util.random_state(spec, K, seed=seed, terminated_p=0.0, density=0.3)
        pg, pa, pt = st["grid"], st["agents"], None
    idx = (np.arange(B) * 3) % K
    env = BatchedMultiGridEnv(spec, B, dev(), first_env=11)
    env.load_state(pg[idx], pa[idx], aux=None if pt is None else pt[idx])
    env.seed_synthetic(seed)
    env.set_layout_pool(pg, pa, pt)
    return env


@pytest.mark.parametrize("name,spec,B,T", AR_CASES, ids=[c[0] for c in AR_CASES])
def test_fused_auto_reset_equals_reset_then_step(name, spec, B, T):
    """mgx_step_autoreset == mgx_reset_done followed by mgx_step, every step, on every output and on the state; and
    mgx_rollout_autoreset == the same sequence in one launch."""
    A = spec.num_agents
    fused, split, roll = _ar_env(spec, B, 21), _ar_env(spec, B, 21), _ar_env(spec, B, 21)
    g = torch.Generator(device=dev()); g.manual_seed(9)
    acts = torch.randint(0, 7, (T, B, A), dtype=torch.int8, device=dev(), generator=g)
    acts[:, ::7, 0] = -1                                            # some agents absent
    r = roll.rollout(acts, auto_reset=True)
    n_reset = 0
    for t in range(T):
        was = split.reset_done().clone()
        want = [x.clone() for x in split.step(acts[t])]
        got = fused.step(acts[t], auto_reset=True)
        assert torch.equal(fused.was_reset, was), f"step {t}: was_reset"
        for k, (gx, wx) in enumerate(zip(got, want)):
            assert torch.equal(gx, wx), f"step {t}: output {k}"
        for k, key in enumerate(("obs", "dir", "reward", "terminated", "truncated")):
            assert torch.equal(r[key][t], want[k]), f"rollout step {t}: {key}"
        assert torch.equal(r["was_reset"][t], was)
        n_reset += int(was.sum())
    for e in (fused, roll):
        for f in ("grid", "agents", "rng", "step_count", "aux", "episode"):
            assert torch.equal(getattr(e, f), getattr(split, f)), f
    assert n_reset > B                                              # every env restarted more than once on average
    fused.check_errors(); split.check_errors(); roll.check_errors()


def test_auto_reset_without_was_reset_output():
    """MgxAutoReset.was_reset is optional (NULL): the state must come out the same."""
    spec = EnvSpec(8, 8, 2, 7, max_steps=5)
    a, b = _ar_env(spec, 300, 3), _ar_env(spec, 300, 3)
    g = torch.Generator(device=dev()); g.manual_seed(2)
    for t in range(14):
        act = torch.randint(0, 7, (300, 2), dtype=torch.int8, device=dev(), generator=g)
        want = [x.clone() for x in a.step(act, auto_reset=True)]
        b.backend.step(b.batch, b.cells, b.agents, b.rng, b.step_count, act, None, b.err, b.obs, b.dir, b.reward,
                       b.terminated, b.truncated, auto_reset=(b.first_env, b._pool, b.episode, None))
        for x, y in zip(want, (b.obs, b.dir, b.reward, b.terminated, b.truncated)):
            assert torch.equal(x, y)
    assert torch.equal(a.grid, b.grid) and torch.equal(a.episode, b.episode) and int(a.episode.sum()) > 300


ROLL_CASES = [
    ("C2_empty16_a4", EnvSpec(16, 16, 4, 7, max_steps=1024), 2048, 40, 0.0),
    ("objects16_a4", EnvSpec(16, 16, 4, 7, max_steps=30), 1500, 40, 0.3),
    ("bup_a2", EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"), 3001, 30, 0.25),
    ("a3_v5_nooverlap_ragged", EnvSpec(9, 7, 3, 5, max_steps=12, allow_agent_overlap=False, failure_termination_mode="any"), 777, 25, 0.3),
    ("a1_v9", EnvSpec(10, 10, 1, 9, max_steps=20), 333, 20, 0.3),
    ("a16_v9_64x64", EnvSpec(64, 64, 16, 9, max_steps=100), 48, 8, 0.1),
]


@pytest.mark.parametrize("name,spec,B,T,density", ROLL_CASES, ids=[c[0] for c in ROLL_CASES])
def test_rollout_equals_repeated_steps(name, spec, B, T, density):
    """mgx_rollout (T steps in one launch, state kept in LDS) == T x mgx_step, bit for bit, incl. final state."""
    st = util.random_state(spec, B, seed=zlib.crc32(name.encode()) % 1000, density=density)
    acts = np.stack([util.random_actions(B, spec.num_agents, seed=300 + t) for t in range(T)])
    e1 = BatchedMultiGridEnv(spec, B, dev()); e1.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    e2 = BatchedMultiGridEnv(spec, B, dev()); e2.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    a = torch.from_numpy(acts).to(dev())
    out = e2.rollout(a)
    for t in range(T):
        obs, dirs, rew, term, trunc = e1.step(a[t])
        ctx = f"{name} step {t}"
        assert torch.equal(out["obs"][t], obs), ctx
        assert torch.equal(out["dir"][t], dirs), ctx
        assert torch.equal(out["reward"][t], rew), ctx
        assert torch.equal(out["terminated"][t], term), ctx
        assert torch.equal(out["truncated"][t], trunc), ctx
    for n in ("grid", "agents", "rng", "step_count"):
        assert torch.equal(getattr(e1, n), getattr(e2, n)), n
    e1.check_errors(); e2.check_errors()
    # and a second rollout continues correctly from the written-back state
    out2 = e2.rollout(a[:3].contiguous())
    for t in range(3):
        obs, *_ = e1.step(a[t])
        assert torch.equal(out2["obs"][t], obs)
    assert torch.equal(e1.grid, e2.grid)


HOOK_CASES = [
    ("redbluedoors6_a3", "MultiGrid-RedBlueDoors-6x6-v0", dict(agents=3, failure_termination_mode="all"), 600, 80),
    ("redbluedoors8_a2", "MultiGrid-RedBlueDoors-8x8-v0", dict(agents=2), 400, 60),
    ("lockedhallway4_a2", "MultiGrid-LockedHallway-4Rooms-v0", dict(agents=2), 300, 60),
    ("playground_a3", "MultiGrid-Playground-v0", dict(agents=3), 200, 60),
]


@pytest.mark.parametrize("name,env_id,kw,B,T", HOOK_CASES, ids=[c[0] for c in HOOK_CASES])
def test_hook_envs_random_rollouts_vs_oracle(name, env_id, kw, B, T):
    """Section 8f-4 envs: B different generated layouts, random actions, every step vs the oracle; also as one rollout."""
    import multigrid_amd as mg
    from multigrid_amd import envs as E
    spec = mg.spec_for(env_id, **kw)
    cls, cfg = mg.CONFIGURATIONS[env_id]
    r = np.random.default_rng(zlib.crc32(name.encode()))
    grids, agents, auxs = [], [], []
    for b in range(B):
        if cls is E.RedBlueDoorsEnv:
            g, a = layouts.redbluedoors_layout(cfg["size"], spec.num_agents, r)
        elif cls is E.LockedHallwayEnv:
            g, a = layouts.lockedhallway_layout(cfg["num_rooms"], 5, 1, 2, spec.num_agents, r, r)
            # hand every agent a key so that doors actually get unlocked by the random walk
            a[:, 5] = 5; a[:, 6] = r.integers(0, 6, size=spec.num_agents)
        else:
            g, a = layouts.playground_layout(7, 3, 3, spec.num_agents, r, r)
        grids.append(g); agents.append(a); auxs.append(layouts.make_aux(spec.env_kind, g))
    st = dict(grid=np.stack(grids), agents=np.stack(agents), aux=np.stack(auxs),
              rng=np.random.default_rng(1).integers(0, 2 ** 63, size=(B, 4), dtype=np.int64).astype(np.uint64) | np.uint64(1),
              step_count=np.zeros(B, np.int32))
    acts = np.stack([np.random.default_rng(50 + t).choice([0, 1, 2, 2, 2, 3, 4, 5, 5, 5, 6], size=(B, spec.num_agents)).astype(np.int8)
                     for t in range(T)])
    env = BatchedMultiGridEnv(spec, B, dev()); env.load_state(st["grid"], st["agents"], st["rng"], st["aux"])
    roll = BatchedMultiGridEnv(spec, B, dev()); roll.load_state(st["grid"], st["agents"], st["rng"], st["aux"])
    out = roll.rollout(torch.from_numpy(acts).to(dev()))
    ref = {k: v.copy() for k, v in st.items()}
    sd = spec.as_dict()
    events = 0
    for t in range(T):
        o_ref, d_ref, r_ref, te_ref, tr_ref = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"],
                                                            acts[t], ref["aux"], nthreads=8)
        obs, dirs, rew, term, trunc = env.step(torch.from_numpy(acts[t]).to(dev()))
        ctx = f"{name} step {t}"
        np.testing.assert_array_equal(env.grid.cpu().numpy(), ref["grid"], err_msg=ctx)
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"], err_msg=ctx)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_ref, err_msg=ctx)
        assert rew.cpu().numpy().tobytes() == r_ref.tobytes(), ctx
        np.testing.assert_array_equal(term.cpu().numpy(), te_ref, err_msg=ctx)
        np.testing.assert_array_equal(trunc.cpu().numpy(), tr_ref, err_msg=ctx)
        if spec.env_kind == "lockedhallway":
            np.testing.assert_array_equal(env.aux.cpu().numpy()[:, [1, 15]], ref["aux"][:, [1, 15]], err_msg=ctx)
        assert torch.equal(out["obs"][t], obs) and torch.equal(out["reward"][t], rew) and torch.equal(out["terminated"][t], term), ctx
        events += int((r_ref > 0).any()) + int(te_ref.any())
    assert torch.equal(env.grid, roll.grid) and torch.equal(env.aux, roll.aux) and torch.equal(env.rng, roll.rng)
    if spec.env_kind != "empty":
        assert events > 0, "the rollout never reached a hook event"
    env.check_errors()


def test_random_specs_soak():
    """A few seconds of tools/fuzz_parity.py: random specs / shapes / states, step and rollout vs the oracle."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "10", "12345"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_captured_graph_of_steps_equals_eager_steps():
    """BatchedMultiGridEnv.capture_steps: a hipGraph of T step launches == T eager steps, and it can be replayed."""
    spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
    B, T = 2048, 12
    st = util.random_state(spec, B, seed=4)
    a = BatchedMultiGridEnv(spec, B, dev()); a.load_state(st["grid"], st["agents"], st["rng"], None, st["step_count"])
    b = BatchedMultiGridEnv(spec, B, dev()); b.load_state(st["grid"], st["agents"], st["rng"], None, st["step_count"])
    acts = torch.from_numpy(np.stack([util.random_actions(B, 4, seed=t) for t in range(T)])).to(dev())
    graph = a.capture_steps(acts)
    for rep in range(2):
        graph.replay()
        for t in range(T):
            want = b.step(acts[t])
        torch.cuda.synchronize()
        for x, y in zip((a.obs, a.dir, a.reward, a.terminated, a.truncated), want):
            assert torch.equal(x, y)
        assert torch.equal(a.grid, b.grid) and torch.equal(a.rng, b.rng) and torch.equal(a.step_count, b.step_count)
    a.check_errors()


@pytest.mark.parametrize("case", ["c4_shape", "bup_pool", "bup_generated", "one_hot_ragged"])
def test_sub_sharded_graph_equals_one_chain(case):
    """capture_steps(sub_shards=P): the batch stepped as P independent chains of launches on P streams (parallel branches of
    one hipGraph) == the same steps as one chain: envs are independent, seeds and auto-reset layouts follow the global env
    index.  Every output of the last step and the whole post-run state are compared, for 2 and 3 sub-shards."""
    from multigrid_amd import workloads
    T = 10
    if case == "c4_shape":
        wl = workloads.make("c4", batch=20000 + 64, first_env=0, global_batch=65536)
        mk = lambda: wl.make_env(dev(), auto_reset=True); kw = dict(auto_reset=True)
    elif case == "bup_pool":
        wl = workloads.make("c3", batch=3000, first_env=128, global_batch=16384)
        mk = lambda: wl.make_env(dev(), auto_reset=True); kw = dict(auto_reset=True)
    elif case == "bup_generated":
        wl = workloads.make("c3", batch=1500, first_env=0, global_batch=16384)

        def mk():
            e = wl.make_env(dev(), auto_reset=False)
            e.set_layout_generator("blockedunlockpickup", layout_seed=5, room_size=6)
            e.step_count.fill_(wl.spec.max_steps - 4)                     # every env is regenerated within the run
            return e
        kw = dict(auto_reset=True)
    else:
        wl = workloads.make("c2", batch=1000 + 7, first_env=0, global_batch=4096)
        mk = lambda: wl.make_env(dev(), auto_reset=True); kw = dict(auto_reset=True, one_hot=True)
    B, A = wl.batch, wl.spec.num_agents
    acts = torch.from_numpy(np.stack([util.random_actions(B, A, seed=300 + t, p_missing=0.0) for t in range(T)])).to(dev())
    ref = mk()
    for t in range(T):
        want = [x.clone() for x in ref.step(acts[t], **kw)]
    for P in (2, 3):
        env = mk()
        shards = env.split(P)
        assert len(shards) == P and sum(s.batch for s in shards) == B and shards[1].first_env == wl.first_env + shards[0].batch
        graph = env.capture_steps(acts, sub_shards=P, **kw)
        graph.replay()
        torch.cuda.synchronize()
        got = ((env._one_hot if kw.get("one_hot") else env.obs), env.dir, env.reward, env.terminated, env.truncated)
        for k, (x, y) in enumerate(zip(got, want)):
            assert torch.equal(x, y), f"{case} P={P}: output {k}"
        for f in ("cells", "agents", "rng", "step_count", "aux", "episode", "was_reset"):
            assert torch.equal(getattr(env, f), getattr(ref, f)), f"{case} P={P}: {f}"
        if case == "bup_generated":
            assert torch.equal(env._gen["gen_state"], ref._gen["gen_state"]) and int(env.episode.sum()) >= B
        env.check_errors()


@pytest.mark.parametrize("kind", ["redbluedoors", "lockedhallway"])
def test_hook_visiting_order_random_vs_oracle(kind):
    """hook_order (the caller's dict order, redbluedoors.py:176 / locked_hallway.py:210): many envs with agents stacked in front
    of a door and toggle-heavy actions, a random permutation per env and step -- HIP vs the oracle, every output and the state;
    the orders must matter (the same run with ascending order gives different results)."""
    from multigrid_amd import layouts
    A, B, T = 3, 600, 12
    r = np.random.default_rng(17)
    grids, agents, auxs = [], [], []
    if kind == "redbluedoors":
        spec = EnvSpec(16, 8, A, 7, max_steps=1280, joint_reward=True, success_termination_mode="any",
                       failure_termination_mode="all", env_kind="redbluedoors")
        for k in range(16):
            g, a = layouts.redbluedoors_layout(8, A, np.random.default_rng(100 + k))
            grids.append(g); agents.append(a); auxs.append(layouts.make_aux("redbluedoors", g))
    else:
        spec = EnvSpec(13, 9, A, 7, max_steps=3200, joint_reward=False, env_kind="lockedhallway")
        for k in range(16):
            g, a = layouts.lockedhallway_layout(4, 5, 1, 2, A, np.random.default_rng(200 + k), np.random.default_rng(300 + k))
            grids.append(g); agents.append(a); auxs.append(layouts.make_aux("lockedhallway", g))
    pick = r.integers(0, 16, size=B)
    grid = np.stack([grids[k] for k in pick]); ag = np.stack([agents[k] for k in pick]); aux = np.stack([auxs[k] for k in pick])
    for b in range(B):                                   # stack every agent in front of a door, facing it; agent 0 holds its key
        ys, xs = np.nonzero(grid[b, :, :, 0] == 4)
        j = r.integers(0, len(xs)); dx_, dy_ = int(xs[j]), int(ys[j])
        # the side of the door the agents stand on: the middle room of RedBlueDoors, the hallway of LockedHallway
        side = (1 if dx_ < 8 else -1) if kind == "redbluedoors" else (1 if dx_ < 6 else -1)
        ag[b, :, 2], ag[b, :, 3], ag[b, :, 1] = dx_ + side, dy_, 2 if side == 1 else 0
        assert grid[b, dy_, dx_ + side, 0] in (1, 5), grid[b, dy_, dx_ + side]
        grid[b, dy_, dx_ + side] = (1, 0, 0)
        if kind == "lockedhallway":
            ag[b, 0, 5:8] = (5, grid[b, dy_, dx_, 1], 0)
    rng = np.random.default_rng(5).integers(0, 2 ** 63, size=(B, 4), dtype=np.int64).astype(np.uint64); rng[:, 2] |= np.uint64(1)
    env = BatchedMultiGridEnv(spec, B, dev())
    env.load_state(grid, ag, rng, aux)
    ref = dict(grid=grid.copy(), agents=ag.copy(), rng=rng.copy(), step_count=np.zeros(B, np.int32), aux=aux.copy())
    asc = {k: v.copy() for k, v in ref.items()}
    sd = spec.as_dict()
    mattered = 0
    for t in range(T):
        act = r.choice([5, 5, 5, 0, 1, 6, -1], size=(B, A)).astype(np.int8)
        order = np.stack([r.permutation(A) for _ in range(B)]).astype(np.uint8)
        want = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["aux"], nthreads=8,
                             hook_order=order)
        plain = ob.step_batch(sd, asc["grid"], asc["agents"], asc["rng"], asc["step_count"], act, asc["aux"], nthreads=8)
        mattered += int(want[2].tobytes() != plain[2].tobytes() or not np.array_equal(want[3], plain[3]))
        got = env.step(torch.from_numpy(act).to(dev()), hook_order=torch.from_numpy(order).to(dev()))
        for k, (g, w) in enumerate(zip(got, want)):
            assert g.cpu().numpy().tobytes() == w.tobytes(), f"{kind} step {t}: output {k}"
        assert env.grid.cpu().numpy().tobytes() == ref["grid"].tobytes()
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"])
        np.testing.assert_array_equal(env.aux.cpu().numpy(), ref["aux"])
    assert mattered >= 1, "the visiting order never mattered: the test is hollow"
    env.check_errors()


def test_layout_tensors_replaced_after_split_or_capture_is_refused():
    """sub-shards and captured graphs hold raw pointers into the layout pool; replacing it must not leave them
    launching on freed memory -- same-shape pools are refilled in place, anything else makes the old objects refuse to run."""
    from multigrid_amd import workloads
    wl = workloads.make("c3", batch=512, first_env=0, global_batch=16384)
    env = wl.make_env(dev(), auto_reset=True)
    acts = torch.from_numpy(np.stack([util.random_actions(512, 2, seed=t, p_missing=0.0) for t in range(3)])).to(dev())
    graph = env.capture_steps(acts, auto_reset=True, sub_shards=2)
    shards = env.split(2)
    pg, pa, pt = wl.pool                                  # numpy: u8[K,H,W,3], u8[K,A,8], u8[K,16]
    env.set_layout_pool(pg, pa, pt)                       # same shapes: in place, everything stays valid
    graph.replay(); shards[0].step(acts[0, :shards[0].batch].contiguous(), auto_reset=True)
    env.set_layout_pool(pg[:100], pa[:100], pt[:100])     # another pool size: new tensors
    with pytest.raises(RuntimeError, match="capture_steps"):
        graph.replay()
    with pytest.raises(RuntimeError, match="split"):
        shards[0].step(acts[0, :shards[0].batch].contiguous(), auto_reset=True)
    env.step(acts[0], auto_reset=True)                    # the env itself re-binds
    torch.cuda.synchronize()
    env.check_errors()
