"""CPU check of multigrid_amd/csrc/mgx_rules.h -- the integer rules the HIP kernels are assembled from
(jump-ahead PCG64, ranking argsort, handle_actions, overlay, hooks, view geometry, in-bounds masks, the closed
form visibility flood) -- against the oracle, via the g++-built host shim.  The kernels' own parallel plumbing
(LDS staging, ballot, byte packing) is covered by the -m gpu parity tests."""
import dataclasses

import numpy as np
import pytest

from multigrid_amd import EnvSpec
from oracle import binding as ob
from tests import hostshim, util

CASES = [
    EnvSpec(16, 16, 4, 7, max_steps=1024),
    EnvSpec(11, 6, 2, 7, max_steps=576, joint_reward=True, env_kind="blockedunlockpickup"),
    EnvSpec(9, 7, 3, 5, max_steps=50, allow_agent_overlap=False, failure_termination_mode="any"),
    EnvSpec(8, 8, 1, 3, max_steps=30, see_through_walls=True),
    EnvSpec(8, 9, 3, 7, max_steps=30, see_through_walls=True),
    EnvSpec(10, 8, 2, 9, max_steps=30, see_through_walls=True),
    EnvSpec(13, 12, 5, 11, max_steps=40, success_termination_mode="all", joint_reward=True),
    EnvSpec(20, 17, 7, 13, max_steps=40),
    EnvSpec(24, 24, 2, 15, max_steps=40),
    EnvSpec(30, 30, 16, 9, max_steps=60),
    EnvSpec(12, 12, 32, 7, max_steps=40),
]


@pytest.mark.parametrize("force_serial", [False, True], ids=["fastpath", "serial"])
@pytest.mark.parametrize("spec", CASES, ids=lambda s: f"{s.width}x{s.height}_a{s.num_agents}_v{s.view_size}")
def test_rules_match_oracle_on_random_states(spec, force_serial):
    B, T = 24, 12
    st = util.random_state(spec, B, seed=spec.width * 100 + spec.num_agents)
    ref = {k: v.copy() for k, v in st.items()}
    sd = spec.as_dict()
    o_ref, _ = ob.gen_obs_batch(sd, ref["grid"], ref["agents"])
    for b in range(B):
        np.testing.assert_array_equal(hostshim.obs_env(spec, st["grid"][b], st["agents"][b]), o_ref[b])
    n_serial = n_total = 0
    for t in range(T):
        act = util.random_actions(B, spec.num_agents, seed=500 + t)
        o_ref, d_ref, r_ref, te_ref, tr_ref = ob.step_batch(
            sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, ref["target"])
        for b in range(B):
            out = hostshim.step_env(spec, st["grid"][b], st["agents"][b], act[b], st["rng"][b],
                                    st["step_count"][b], st["target"][b], force_serial)
            n_serial += out["serial"]; n_total += 1
            st["step_count"][b] = out["step_count"]
            ctx = f"step {t} env {b}"
            assert out["rc"] == 0
            np.testing.assert_array_equal(st["grid"][b], ref["grid"][b], err_msg=ctx)
            np.testing.assert_array_equal(st["agents"][b], ref["agents"][b], err_msg=ctx)
            np.testing.assert_array_equal(out["obs"], o_ref[b], err_msg=ctx)
            assert out["reward"].tobytes() == r_ref[b].tobytes(), ctx
            np.testing.assert_array_equal(out["terminated"], te_ref[b], err_msg=ctx)
            assert out["truncated"] == tr_ref[b], ctx
            if spec.num_agents > 1:
                np.testing.assert_array_equal(st["rng"][b], ref["rng"][b], err_msg=ctx)
    if not force_serial and spec.num_agents <= 7:
        assert n_serial < n_total, (n_serial, n_total)          # the order-free path is exercised ...
        if spec.num_agents >= 5 or not spec.allow_agent_overlap:
            assert n_serial > 0, (n_serial, n_total)            # ... and so is the fallback (cell conflicts, presence)


COMPACT = [dataclasses.replace(s, cell_bytes=1) for s in (CASES[0], CASES[1], CASES[2], CASES[6], CASES[9])] + [
    EnvSpec(8, 8, 2, 7, max_steps=40, env_kind="redbluedoors", cell_bytes=1)]


@pytest.mark.parametrize("force_serial", [False, True], ids=["fastpath", "serial"])
@pytest.mark.parametrize("spec", COMPACT, ids=lambda s: f"{s.width}x{s.height}_a{s.num_agents}_v{s.view_size}_{s.env_kind}")
def test_rules_on_compact_cells_match_oracle(spec, force_serial):
    """The same rules on COMPACT one-byte cells (include/mgx.h: MgxCell8, EnvSpec.cell_bytes = 1: type and state coded jointly):
    the host shim packs the tile that way, the rules read / write it through the format-aware accessors, and every output and the
    whole post-step state must equal the oracle's -- which knows nothing about cell formats."""
    if spec.env_kind == "redbluedoors":
        pytest.skip("random states of the hook envs come from their own generators (tests/test_hip_parity.py on the GPU)")
    test_rules_match_oracle_on_random_states(spec, force_serial)


def test_compact_cell_codes_round_trip():
    from multigrid_amd import layouts
    cells = [(t, c, 0) for t in range(11) for c in range(6)] + [(4, c, s) for c in range(6) for s in (1, 2)] \
        + [(10, c, d) for c in range(6) for d in (1, 2, 3)]
    g = np.array(cells, dtype=np.uint8)
    p = layouts.pack_cells8(g)
    assert len(set(p.tolist())) == len(cells)                         # distinct codes
    np.testing.assert_array_equal(layouts.unpack_cells8(p), g)
    assert p[cells.index((2, 5, 0))] == 0xD2                           # WALL
    opaque = np.array([(t == 2) or (t == 4 and s != 0) for t, c, s in cells])
    np.testing.assert_array_equal((p & 0x80) != 0, opaque)
    with pytest.raises(ValueError):
        layouts.pack_cells8(np.array([[5, 1, 1]], dtype=np.uint8))    # a state on a key: not representable


@pytest.mark.parametrize("path", util.GOLDEN, ids=util.GOLDEN_IDS)
def test_rules_replay_goldens(path):
    from multigrid_amd import layouts
    z, d, spec = util.load_golden(path)
    tile = layouts.grid_to_product(z["grid0"]); rows = layouts.pack_agents(z["agents0"])
    rng = util.rng_words_lohi(z["rng0"]); target = util.golden_target(d); sc = 0
    np.testing.assert_array_equal(hostshim.obs_env(spec, tile, rows), z["obs0"])
    for t in range(z["actions"].shape[0]):
        out = hostshim.step_env(spec, tile, rows, np.ascontiguousarray(z["actions"][t]), rng, sc, target,
                                hook_order=z["hook_order"][t] if "hook_order" in z.files else None)
        sc = out["step_count"]
        ctx = f"step {t}"
        np.testing.assert_array_equal(out["order"][:spec.num_agents] if spec.num_agents > 1 else [0],
                                      z["order"][t], err_msg=ctx)
        np.testing.assert_array_equal(out["obs"], z["obs"][t], err_msg=ctx)
        assert out["reward"].tobytes() == z["reward"][t].tobytes(), ctx
        np.testing.assert_array_equal(out["terminated"], z["terminated"][t], err_msg=ctx)
        assert out["truncated"] == int(z["truncated"][t]), ctx
        np.testing.assert_array_equal(layouts.grid_from_product(tile), z["grid"][t].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(layouts.unpack_agents(rows), z["agents"][t].astype(np.int64), err_msg=ctx)
    np.testing.assert_array_equal(rng, util.rng_words_lohi(z["rng_final"]))


def test_pool_index_by_multiplication_is_the_exact_remainder():
    """The restart's layout index (mgx_rules.h: pool_index) takes three 32-bit remainders by multiplication when the pool has at
    most 2^19 layouts and the env index is below 2^32, the 64-bit division otherwise: both must equal the definition of
    include/mgx.h (MgxAutoReset), (first_env + b + episode * 7919) mod K, computed here in Python integers."""
    import ctypes as C
    L = hostshim.lib()
    L.shim_pool_index.restype = C.c_int
    L.shim_pool_index.argtypes = [C.c_int64, C.c_int64, C.c_int32, C.c_int32]
    r = np.random.default_rng(11)
    ks = [1, 2, 3, 7, 64, 255, 256, 1000, 7918, 7919, 7920, 65535, 65536, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, (1 << 31) - 1]
    ks += [int(x) for x in r.integers(1, 1 << 19, 40)] + [int(x) for x in r.integers(1 << 19, (1 << 31) - 1, 10)]
    cases = 0
    for K in ks:
        firsts = [0, 1, K - 1, K, (1 << 32) - 2, (1 << 32) - 1, 1 << 32, (1 << 40) + 5] + [int(x) for x in r.integers(0, 1 << 33, 6)]
        for first in firsts:
            for b in (0, 1, 63, 65535, int(r.integers(0, 1 << 20))):
                for ep in (0, 1, 2, 542_000, (1 << 31) - 1, int(r.integers(0, 1 << 31))):
                    assert L.shim_pool_index(first, b, ep, K) == (first + b + ep * 7919) % K, (first, b, ep, K)
                    cases += 1
    assert cases > 20000


def test_pcg64_advance_equals_that_many_numpy_draws():
    """mgx_rules.h: pcg64_advance -- what the staged generator uses to carry a snapshot of env.np_random `lead` steps forward (A draws
    per step, multigrid/base.py:399) -- against numpy's own generator advanced draw by draw."""
    import ctypes as C
    L = hostshim.lib()
    L.shim_pcg64_advance.restype = None
    L.shim_pcg64_advance.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.c_uint32]
    m = (1 << 64) - 1
    for seed, k, n in [(0, 2, 64), (1, 2, 128), (2, 4, 1), (3, 4, 127), (4, 3, 1000), (5, 16, 575), (6, 1, 0), (7, 32, 4097)]:
        g = np.random.Generator(np.random.PCG64(seed))
        st = g.bit_generator.state["state"]
        w = (C.c_uint64 * 4)(st["state"] & m, st["state"] >> 64, st["inc"] & m, st["inc"] >> 64)
        L.shim_pcg64_advance(w, k, n)
        if n * k:
            g.random(n * k)
        st = g.bit_generator.state["state"]
        assert (int(w[0]), int(w[1]), int(w[2]), int(w[3])) == (st["state"] & m, st["state"] >> 64, st["inc"] & m, st["inc"] >> 64), (seed, k, n)
