"""Every instantiation the library ships is LAUNCHED by this suite ("an instantiation nobody launched is untested code on
the product path").  The fused kernel is a template over (view size, mode, hooks, auto-reset, one-hot, generation, streamed tile loads,
LDS-DMA, shape, compact cells, byte grids) and the launcher picks among ~500 instantiations by spec, batch and flags; the other test
files land in the families they were written for.  This one walks the matrix: for every view size 3..15 and every launch regime --

    latency     <= 2048 wavefronts, 16-bit cells: the LDS-DMA instantiations
    throughput  > 2048 wavefronts: 64 / 32 view slots, tile through registers
    streamed    a grid tensor beyond 128 MiB: non-temporal tile loads
    compact / byte-grid cells (their own families, at any batch)

-- x {hook env, hook-free} x {auto-reset from a layout pool, none} x {3-channel, one-hot observations}: gen_obs, the step, the rollout
and the persistent launch, each against `BatchedMultiGridEnv` on the CPU-oracle backend (tests/util.OracleBackend: the oracle's step
behind the same host logic) on the same random states -- every output of every step and the state afterwards.  The big batches of the
streamed cases are compared on slices of the batch (the kernels are data-parallel over envs; the oracle then costs milliseconds).

tools/kernel_coverage.py records which kernels a run of `pytest -m gpu` launched (profiles/kernels_launched.txt) and
tests/test_kernel_coverage.py (CPU) fails when the library carries a kernel that is not on that list.
Reference semantics: multigrid/base.py:303-376 (step, gen_obs), utils/obs.py:65-102, wrappers.py:158-190 (one-hot)."""
import dataclasses
import functools

import numpy as np
import pytest
import torch

from multigrid_amd import BatchedMultiGridEnv, EnvSpec
from oracle import binding as ob
from tests import util

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
VIEWS = (3, 5, 7, 9, 11, 13, 15)
NT = ob.max_threads()


@functools.lru_cache(maxsize=4)
def _state(W, H, A, B, seed):
    """Random states for B envs (view-independent: shared by every view size and variant of a family).  Big batches repeat a block of
    2048 distinct envs: what is compared against the oracle are slices of the batch anyway."""
    spec = EnvSpec(W, H, A, 7, max_steps=40)
    n = min(B, 2048)
    st = util.random_state(spec, n, seed=seed, density=0.2, terminated_p=0.15)
    st["step_count"][::3] = 39                                    # a third of the envs truncate with the first step
    if n < B:
        st = {k: np.concatenate([v] * (-(-B // n)))[:B] for k, v in st.items()}
    pool = util.random_state(spec, 3, seed=seed + 1, density=0.25, terminated_p=0.0)
    return st, pool


def _spec(W, H, A, V, hooks, cell_bytes):
    return EnvSpec(W, H, A, V, max_steps=40, joint_reward=hooks, env_kind="blockedunlockpickup" if hooks else "empty",
                   cell_bytes=cell_bytes)


def _envs(W, H, A, V, B, hooks, cell_bytes, auto_reset, sample=None):
    """(hip env over all B envs, oracle-backend env over `sample` (a slice list) of them or all)."""
    st, pool = _state(W, H, A, B, 1000 + W + A)
    spec = _spec(W, H, A, V, hooks, cell_bytes)
    hip = BatchedMultiGridEnv(spec, B, DEV, first_env=5)
    hip.load_state(st["grid"], st["agents"], st["rng"], st["target"] if hooks else None, st["step_count"], validate=False)
    idx = np.arange(B) if sample is None else np.concatenate([np.arange(lo, hi) for lo, hi in sample])
    rspec = dataclasses.replace(spec, cell_bytes=2)               # (the oracle knows nothing about cell formats)
    ref = [BatchedMultiGridEnv(rspec, hi - lo, "cpu", first_env=5 + lo, backend=util.OracleBackend(rspec, nthreads=NT))
           for lo, hi in ([(0, B)] if sample is None else sample)]
    for r, (lo, hi) in zip(ref, [(0, B)] if sample is None else sample):
        r.load_state(st["grid"][lo:hi], st["agents"][lo:hi], st["rng"][lo:hi], st["target"][lo:hi] if hooks else None,
                     st["step_count"][lo:hi], validate=False)
    if auto_reset:
        for e in [hip] + ref:
            e.set_layout_pool(pool["grid"], pool["agents"], pool["target"] if hooks else None)
    return hip, ref, idx, st


def _ref_step(ref, acts, auto_reset, one_hot, sample, B):
    outs = []
    for r, (lo, hi) in zip(ref, [(0, B)] if sample is None else sample):
        if auto_reset:
            r.reset_done()
        o = r.step(torch.from_numpy(acts[lo:hi]), one_hot=one_hot)
        outs.append([x.numpy().copy() for x in o] + ([r.was_reset.numpy().copy()] if auto_reset else []))
    return [np.concatenate([o[k] for o in outs]) for k in range(len(outs[0]))]


def _check_state(hip, ref, idx, ctx):
    for name in ("grid", "agents", "step_count"):
        want = np.concatenate([getattr(r, name).numpy() for r in ref])
        np.testing.assert_array_equal(getattr(hip, name).cpu().numpy()[idx], want, err_msg=f"{ctx}: {name}")
    if hip.spec.num_agents > 1:
        want = np.concatenate([r.rng.numpy() for r in ref])
        np.testing.assert_array_equal(hip.rng.cpu().numpy()[idx], want, err_msg=f"{ctx}: rng")


def _steps(W, H, A, V, B, hooks, cell_bytes, auto_reset, one_hot, sample=None, T=2, ctx=""):
    hip, ref, idx, st = _envs(W, H, A, V, B, hooks, cell_bytes, auto_reset, sample)
    for t in range(T):
        acts = util.random_actions(B, A, seed=50 + t)
        got = hip.step(torch.from_numpy(acts).to(DEV), auto_reset=auto_reset, one_hot=one_hot)
        want = _ref_step(ref, acts, auto_reset, one_hot, sample, B)
        got = [g.cpu().numpy()[idx] for g in got] + ([hip.was_reset.cpu().numpy()[idx]] if auto_reset else [])
        for k, (g, w) in enumerate(zip(got, want)):
            assert g.tobytes() == w.tobytes(), f"{ctx} step {t}: output {k}"
    _check_state(hip, ref, idx, ctx)
    hip.check_errors()
    return hip


def _gen_obs(W, H, A, V, B, cell_bytes, one_hot, sample=None, ctx=""):
    hip, ref, idx, st = _envs(W, H, A, V, B, False, cell_bytes, False, sample)
    got, gd = hip.gen_obs(one_hot=one_hot)
    want = np.concatenate([r.gen_obs()[0].numpy() for r in ref])
    if one_hot:
        want = ob.one_hot(want)
    assert got.cpu().numpy()[idx].tobytes() == want.tobytes(), ctx
    np.testing.assert_array_equal(gd.cpu().numpy()[idx], np.concatenate([r.dir.numpy() for r in ref]), err_msg=ctx)


def _waves(spec, B):
    li = BatchedMultiGridEnv(spec, 1, DEV).backend.launch_info(B)
    return -(-B // li["envs_per_wavefront"])


# (hooks, auto_reset) of the step kernels
HA = [(h, a) for h in (False, True) for a in (False, True)]


@pytest.mark.parametrize("V", VIEWS)
def test_latency_family(V):
    """<= 2048 wavefronts on 16-bit cells: the LDS-DMA step instantiations (one-hot output has none: those launches take the
    throughput kernels at any size -- covered here too), gen_obs, the rollout and the persistent launch."""
    W, H, A, B = 12, 10, 2, 300
    assert _waves(_spec(W, H, A, V, False, 2), B) <= 2048
    for hooks, ar in HA:
        _steps(W, H, A, V, B, hooks, 2, ar, False, ctx=f"latency v{V} hooks={hooks} ar={ar}")
        _steps(W, H, A, V, B, hooks, 2, ar, True, ctx=f"latency v{V} one-hot hooks={hooks} ar={ar}")
    _gen_obs(W, H, A, V, B, 2, False, ctx=f"latency gen_obs v{V}")
    _gen_obs(W, H, A, V, B, 2, True, ctx=f"latency gen_obs one-hot v{V}")


@pytest.mark.parametrize("V", VIEWS)
def test_rollout_and_persistent_family(V):
    """mgx_rollout* (x one-hot) and mgx_step_persistent, x hooks x auto-reset, against repeated oracle steps."""
    W, H, A, B, T = 12, 10, 2, 300, 3
    for hooks, ar in HA:
        for one_hot in (False, True):
            hip, ref, idx, st = _envs(W, H, A, V, B, hooks, 2, ar)
            acts = np.stack([util.random_actions(B, A, seed=70 + t) for t in range(T)])
            out = hip.rollout(torch.from_numpy(acts).to(DEV), auto_reset=ar, one_hot=one_hot)
            for t in range(T):
                want = _ref_step(ref, acts[t], ar, one_hot, None, B)
                got = [out[k][t].cpu().numpy() for k in ("obs", "dir", "reward", "terminated", "truncated")] \
                    + ([out["was_reset"][t].cpu().numpy()] if ar else [])
                for k, (g, w) in enumerate(zip(got, want)):
                    assert g.tobytes() == w.tobytes(), f"rollout v{V} hooks={hooks} ar={ar} one_hot={one_hot} step {t}: output {k}"
            _check_state(hip, ref, idx, f"rollout v{V}")
        hip, ref, idx, st = _envs(W, H, A, V, B, hooks, 2, ar)
        with hip.persistent(max_steps=T, auto_reset=ar) as ps:
            for t in range(T):
                acts = util.random_actions(B, A, seed=90 + t)
                got = [g.cpu().numpy() for g in ps.step(torch.from_numpy(acts).to(DEV))] + ([hip.was_reset.cpu().numpy()] if ar else [])
                want = _ref_step(ref, acts, ar, False, None, B)
                for k, (g, w) in enumerate(zip(got, want)):
                    assert g.tobytes() == w.tobytes(), f"persistent v{V} hooks={hooks} ar={ar} step {t}: output {k}"
        assert ps.timeouts == 0
        _check_state(hip, ref, idx, f"persistent v{V}")
    for ar in (False, True):                                   # compact cells: the hook-free rollout and persistent launch (round 6)
        hip, ref, idx, st = _envs(W, H, A, V, B, False, 1, ar)
        acts = np.stack([util.random_actions(B, A, seed=70 + t) for t in range(T)])
        out = hip.rollout(torch.from_numpy(acts).to(DEV), auto_reset=ar)
        for t in range(T):
            want = _ref_step(ref, acts[t], ar, False, None, B)
            got = [out[k][t].cpu().numpy() for k in ("obs", "dir", "reward", "terminated", "truncated")] + ([out["was_reset"][t].cpu().numpy()] if ar else [])
            for k, (g, w) in enumerate(zip(got, want)):
                assert g.tobytes() == w.tobytes(), f"compact rollout v{V} ar={ar} step {t}: output {k}"
        hip, ref, idx, st = _envs(W, H, A, V, B, False, 1, ar)
        with hip.persistent(max_steps=T, auto_reset=ar) as ps:
            for t in range(T):
                acts1 = util.random_actions(B, A, seed=90 + t)
                got = [g.cpu().numpy() for g in ps.step(torch.from_numpy(acts1).to(DEV))] + ([hip.was_reset.cpu().numpy()] if ar else [])
                want = _ref_step(ref, acts1, ar, False, None, B)
                for k, (g, w) in enumerate(zip(got, want)):
                    assert g.tobytes() == w.tobytes(), f"compact persistent v{V} ar={ar} step {t}: output {k}"
        assert ps.timeouts == 0
        _check_state(hip, ref, idx, f"compact persistent v{V}")


@pytest.mark.parametrize("V", VIEWS)
def test_throughput_family(V):
    """> 2048 wavefronts on 16-bit cells: 64 view slots (views up to 7x7) / 32, the tile through registers."""
    W, H, A, B = 16, 12, 16, 8250
    assert _waves(_spec(W, H, A, V, False, 2), B) > 2048
    sample = [(0, 200), (4000, 4200), (B - 200, B)]
    for hooks, ar in HA:
        _steps(W, H, A, V, B, hooks, 2, ar, False, sample, ctx=f"throughput v{V} hooks={hooks} ar={ar}")
    _gen_obs(W, H, A, V, B, 2, False, sample, ctx=f"throughput gen_obs v{V}")
    _gen_obs(W, H, A, V, B, 2, True, sample, ctx=f"throughput gen_obs one-hot v{V}")


@pytest.mark.parametrize("V", VIEWS)
def test_compact_and_byte_grid_families(V):
    """EnvSpec.cell_bytes = 1 (compact cells) and 3 (the reference's byte triples) at a batch the caches hold."""
    W, H, A, B = 12, 10, 2, 300
    for cb in (1, 3):
        for hooks, ar in HA:
            _steps(W, H, A, V, B, hooks, cb, ar, False, ctx=f"cell_bytes={cb} v{V} hooks={hooks} ar={ar}")
        _gen_obs(W, H, A, V, B, cb, False, ctx=f"cell_bytes={cb} gen_obs v{V}")
    for ar in (False, True):                                   # compact cells: the hook-free step with one-hot output (round 6)
        _steps(W, H, A, V, B, False, 1, ar, True, ctx=f"compact one-hot v{V} ar={ar}")


@pytest.mark.parametrize("V", VIEWS)
def test_streamed_family(V):
    """A grid tensor beyond 128 MiB: non-temporal tile loads, every cell format.  64x64 grids; the oracle sees three slices."""
    W = H = 64
    A = 2
    for cb, B in ((2, 16500), (1, 32900)):
        nbytes = B * W * H * cb
        assert nbytes > (128 << 20)
        sample = [(0, 64), (B // 2, B // 2 + 64), (B - 64, B)]
        for hooks, ar in HA:
            _steps(W, H, A, V, B, hooks, cb, ar, False, sample, T=1, ctx=f"streamed cell_bytes={cb} v{V} hooks={hooks} ar={ar}")
            if cb == 2 or not hooks:
                _steps(W, H, A, V, B, hooks, cb, ar, True, sample, T=1, ctx=f"streamed one-hot cell_bytes={cb} v{V} hooks={hooks} ar={ar}")
        _gen_obs(W, H, A, V, B, cb, False, sample, ctx=f"streamed gen_obs cell_bytes={cb} v{V}")
        if cb == 2:
            _gen_obs(W, H, A, V, B, cb, True, sample, ctx=f"streamed gen_obs one-hot v{V}")
    torch.cuda.empty_cache()


@pytest.mark.parametrize("V", VIEWS)
def test_generated_step_family(V):
    """mgx_step_generate (the episodes that end are regenerated in the step's own launch) x hooks x one-hot, every view size: a
    hook-free generator (Empty-Random) and BlockedUnlockPickup, against the oracle backend's step + reset_generate."""
    from tests.test_layout_gen import _make
    for spec, gen in ((EnvSpec(7, 7, 3, V, max_steps=4), dict(kind="empty_random")),
                      (EnvSpec(11, 6, 2, V, max_steps=5, joint_reward=True, env_kind="blockedunlockpickup"),
                       dict(kind="blockedunlockpickup", room_size=6))):
        for one_hot in (False, True):
            B = 130
            hip, ref = _make(spec, gen, B, DEV), _make(spec, gen, B, "cpu", backend=util.OracleBackend(spec, nthreads=8))
            for t in range(2 * spec.max_steps + 2):
                act = torch.from_numpy(util.random_actions(B, spec.num_agents, seed=t, p_missing=0.0))
                got = hip.step(act.to(DEV), auto_reset=True, one_hot=one_hot)
                want = [x.clone() for x in ref.step(act, one_hot=one_hot)]
                ref.reset_done()
                for k, (g, w) in enumerate(zip(got, want)):
                    assert g.cpu().numpy().tobytes() == w.numpy().tobytes(), f"generate v{V} {gen['kind']} one_hot={one_hot} step {t}: {k}"
                assert torch.equal(hip.was_reset.cpu(), ref.was_reset), f"generate v{V} step {t}: was_reset"
            for f in ("grid", "agents", "rng", "step_count", "episode"):
                assert torch.equal(getattr(hip, f).cpu(), getattr(ref, f)), f
            assert int(hip.episode.sum()) >= B


SHAPED = [
    # workload, batch, cell_bytes, kShapes entry of the step, persistent too?
    ("c2", 4096, 2, 1, True), ("c4", 16384, 2, 2, True), ("c3", 16384, 2, 3, True),
    ("c5", 32768, 2, 4, False), ("c5", 36864, 1, 5, False), ("c5", 32768, 1, 6, False),
]


@pytest.mark.parametrize("name,B,cb,shape,persist", SHAPED, ids=[f"shape{c[3]}" for c in SHAPED])
def test_shape_specialised_kernels_with_and_without_auto_reset(name, B, cb, shape, persist):
    """The shape-specialised instantiations (mgx_fused.h: kShapes) of the step -- and, for the latency shapes, of the persistent
    launch -- WITHOUT the fused auto-reset as well (the bench and tests/test_full_size.py run the auto-reset forms)."""
    from multigrid_amd import workloads
    G = max(B, workloads.GLOBAL_BATCH[name])
    wl = workloads.make(name, batch=B, global_batch=G, cell_bytes=cb)
    sample = [(0, 64), (B // 2, B // 2 + 64), (B - 64, B)]
    rspec = dataclasses.replace(wl.spec, cell_bytes=2)
    for ar in (False, True):
        hip = wl.make_env(DEV, auto_reset=ar)
        assert hip.backend.launch_info(B)["fixed_shape"] == shape
        refs = []
        for lo, hi in sample:
            w = workloads.make(name, batch=hi - lo, first_env=lo, global_batch=G, cell_bytes=2)
            refs.append(w.make_env("cpu", backend=util.OracleBackend(rspec, nthreads=NT), auto_reset=ar))
        for e in [hip] + refs:
            e.step_count.fill_(wl.spec.max_steps - 1)            # (with auto-reset: every env restarts at the second step)
        idx = np.concatenate([np.arange(lo, hi) for lo, hi in sample])
        sess = hip.persistent(max_steps=3, auto_reset=ar) if persist else None
        if sess is not None:
            sess.__enter__()
        for t in range(3):
            acts = util.random_actions(B, wl.spec.num_agents, seed=400 + t, p_missing=0.0)
            a = torch.from_numpy(acts).to(DEV)
            got = sess.step(a) if sess is not None else hip.step(a, auto_reset=ar)
            want = _ref_step(refs, acts, ar, False, sample, B)
            for k, (g, w) in enumerate(zip(got, want)):
                assert g.cpu().numpy()[idx].tobytes() == w.tobytes(), f"shape {shape} ar={ar} persist={sess is not None} step {t}: output {k}"
        if sess is not None:
            sess.close()
            assert sess.timeouts == 0
            hip2 = wl.make_env(DEV, auto_reset=ar)                    # ... and the plain step of the same shape
            hip2.step_count.fill_(wl.spec.max_steps - 1)
            refs2 = []
            for lo, hi in sample:
                w = workloads.make(name, batch=hi - lo, first_env=lo, global_batch=G, cell_bytes=2)
                refs2.append(w.make_env("cpu", backend=util.OracleBackend(rspec, nthreads=NT), auto_reset=ar))
                refs2[-1].step_count.fill_(wl.spec.max_steps - 1)
            for t in range(2):
                acts = util.random_actions(B, wl.spec.num_agents, seed=500 + t, p_missing=0.0)
                got = hip2.step(torch.from_numpy(acts).to(DEV), auto_reset=ar)
                want = _ref_step(refs2, acts, ar, False, sample, B)
                for k, (g, w) in enumerate(zip(got, want)):
                    assert g.cpu().numpy()[idx].tobytes() == w.tobytes(), f"shape {shape} ar={ar} step {t}: output {k}"
        _check_state(hip, refs, idx, f"shape {shape} ar={ar}")


def test_resident_shapes_of_the_persistent_launch_with_and_without_auto_reset(monkeypatch):
    """kShapes 7 / 8 of the persistent launch, forced at a small batch, x auto-reset (tests/test_resident.py: the rollouts, the full size)."""
    spec = EnvSpec(16, 16, 4, 7, max_steps=40)
    for ns in (1, 2):
        for ar in (False, True):
            B, T = 80, 3
            hip, ref, idx, st = _envs(16, 16, 4, 7, B, False, 2, ar)
            monkeypatch.setenv("MGX_RESIDENT_SHAPE", {1: "7", 2: "8", 9: "9"}[ns])
            with hip.persistent(max_steps=T, auto_reset=ar) as ps:
                for t in range(T):
                    acts = util.random_actions(B, 4, seed=t)
                    got = [g.cpu().numpy() for g in ps.step(torch.from_numpy(acts).to(DEV))] + ([hip.was_reset.cpu().numpy()] if ar else [])
                    want = _ref_step(ref, acts, ar, False, None, B)
                    for g, w in zip(got, want):
                        assert g.tobytes() == w.tobytes(), f"persistent ns={ns} ar={ar} step {t}"
            monkeypatch.setenv("MGX_RESIDENT_SHAPE", "")
            assert ps.timeouts == 0
            _check_state(hip, ref, idx, f"persistent ns={ns} ar={ar}")


def test_generated_step_at_throughput_batches():
    """mgx_step_generate beyond 2048 wavefronts at 7x7 views (smaller batches of that view size take the LDS-DMA instantiation):
    hook-free (Empty-Random) and BlockedUnlockPickup; the oracle backend steps the first and the last envs of the batch."""
    from tests.test_layout_gen import _make
    for spec, gen, B in ((EnvSpec(9, 9, 16, 7, max_steps=3), dict(kind="empty_random"), 8400),
                         (EnvSpec(11, 6, 2, 7, max_steps=4, joint_reward=True, env_kind="blockedunlockpickup"),
                          dict(kind="blockedunlockpickup", room_size=6), 66000)):
        hip = _make(spec, gen, B, DEV)
        assert -(-B // hip.backend.launch_info(B)["envs_per_wavefront"]) > 2048
        n = 48
        refs = []
        for lo in (0, B - n):
            r = _make(spec, gen, n, "cpu", backend=util.OracleBackend(spec, nthreads=8))
            refs.append((lo, r))
        # (the shards of one global job: seeds, generator streams and layouts are functions of the GLOBAL env index)
        for lo, r in refs:
            r.first_env = hip.first_env + lo
            r.seed_synthetic(3)
            r.set_layout_generator(layout_seed=11, **gen)
        for t in range(2 * spec.max_steps + 1):
            acts = util.random_actions(B, spec.num_agents, seed=t, p_missing=0.0)
            got = hip.step(torch.from_numpy(acts).to(DEV), auto_reset=True)
            for lo, r in refs:
                want = [x.clone() for x in r.step(torch.from_numpy(acts[lo:lo + n]))]
                r.reset_done()
                for k, (g, w) in enumerate(zip(got, want)):
                    assert g[lo:lo + n].cpu().numpy().tobytes() == w.numpy().tobytes(), f"{gen['kind']} B={B} step {t} envs {lo}..: {k}"
                assert torch.equal(hip.was_reset[lo:lo + n].cpu(), r.was_reset)
        for lo, r in refs:
            for f in ("grid", "agents", "rng", "step_count", "episode"):
                assert torch.equal(getattr(hip, f)[lo:lo + n].cpu(), getattr(r, f)), f


def test_the_small_kernels_beside_the_fused_one():
    """full_obs on byte grids, the persistent feeder for agent counts that are not multiples of four, auto-reset layouts whose byte
    count is odd / a multiple of 8 only (reset_done_kernel's copy units)."""
    # (9x7x3 on byte grids: full_obs_kernel<3>; 3 agents: persistent_feed_kernel<0>; 1 / 2: <1> / <2>; 6x6: a 72-byte layout, copied
    # in 8-byte units; 9x7: 126 bytes, 2-byte units)
    for (W, H, A, cb) in ((9, 7, 3, 3), (9, 7, 3, 2), (9, 7, 1, 2), (11, 6, 2, 2), (6, 6, 2, 2), (12, 12, 4, 2)):
        spec = EnvSpec(W, H, A, 5, max_steps=30, cell_bytes=cb)
        st = util.random_state(spec, 200, seed=3)
        hip = BatchedMultiGridEnv(spec, 200, DEV)
        rspec = dataclasses.replace(spec, cell_bytes=2)
        ref = BatchedMultiGridEnv(rspec, 200, "cpu", backend=util.OracleBackend(rspec))
        for e in (hip, ref):
            e.load_state(st["grid"], st["agents"], st["rng"], None, st["step_count"], validate=False)
        np.testing.assert_array_equal(hip.full_obs().cpu().numpy(), ref.full_obs().numpy())
        pool = util.random_state(spec, 2, seed=4, terminated_p=0.0)
        for e in (hip, ref):
            e.set_layout_pool(pool["grid"], pool["agents"])
            e.step_count.fill_(30)
            e.reset_done()
        for name in ("grid", "agents", "step_count", "episode"):       # (agents: a single agent's row once went to the wrong env)
            np.testing.assert_array_equal(getattr(hip, name).cpu().numpy(), getattr(ref, name).numpy(), err_msg=f"{W}x{H} A={A}: {name}")
        if cb == 2:
            T = 6
            acts = torch.from_numpy(np.stack([util.random_actions(200, A, seed=t, p_missing=0.0) for t in range(T)])).to(DEV)
            with hip.persistent(max_steps=T) as ps:
                ps.feed(acts)
            assert ps.timeouts == 0 and ps.steps_completed == T
            for t in range(T):
                want = ref.step(acts[t].cpu())
            for w, g in zip(want, (hip.obs, hip.dir, hip.reward, hip.terminated, hip.truncated)):
                assert g.cpu().numpy().tobytes() == w.numpy().tobytes()
