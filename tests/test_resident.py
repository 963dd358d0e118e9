"""The RESIDENT forms of the C4 shape (round 6; mgx_fused.h: kShapes 7 / 8): mgx_rollout* and mgx_step_persistent for
Empty-16x16 x 4 agents with 64 view slots per wavefront and one or TWO slices of 16 envs per wavefront -- 2048 wavefronts keep the
65536 envs of BASELINE.json configs[3], tiles and all, on the chip for the whole launch.  Same results as T x mgx_step, bit for bit:
every output of every step and the state written back at the end; at the full size also against the CPU oracle.
Reference semantics: multigrid/base.py:303-346 (step), utils/obs.py:65-102 (gen_obs)."""
import numpy as np
import pytest
import torch

from multigrid_amd import _lib, workloads
from multigrid_amd.batched import BatchedMultiGridEnv
from multigrid_amd.spec import EnvSpec
from oracle import binding as ob

from . import util
from .test_full_size import oracle_reset_done

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(spec, B, seed, density):
    st = util.random_state(spec, B, seed=seed, density=density)
    envs = []
    for _ in range(2):
        e = BatchedMultiGridEnv(spec, B, DEV)
        e.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
        envs.append(e)
    return envs


def _acts(B, A, T, seed):
    return torch.from_numpy(np.stack([util.random_actions(B, A, seed=seed + t) for t in range(T)])).to(DEV)


CASES = [
    # slices, batch, steps, max_steps, object density
    (1, 16, 12, 1024, 0.0),
    (1, 37, 40, 30, 0.3),          # ragged: the last wavefront holds 5 envs
    (1, 1500, 40, 30, 0.3),
    (2, 16, 12, 1024, 0.0),        # one wavefront whose second slice is empty
    (2, 37, 40, 30, 0.3),          # wavefront 0: two slices, wavefront 1: 5 envs in its first
    (2, 53, 40, 30, 0.3),          # ... 16 + 5
    (2, 1500, 40, 30, 0.3),
    (2, 4096 + 24, 10, 1024, 0.25),
]


CASES += [(9, b, t, m, d) for (n, b, t, m, d) in CASES if n == 1]      # kShapes 9: the ring-sharing tile, 16 wavefronts per CU


def _force(monkeypatch, ns):
    """ns = 1 / 2: kShapes 7 / 8 (MGX_RESIDENT_SLICES); 9: kShapes 9; 0: none."""
    monkeypatch.setenv("MGX_RESIDENT_SHAPE", "9" if ns == 9 else "")
    monkeypatch.setenv("MGX_RESIDENT_SLICES", "" if ns == 9 else str(ns))


@pytest.mark.parametrize("ns,B,T,max_steps,density", CASES, ids=[f"ns{c[0]}_b{c[1]}" for c in CASES])
def test_resident_rollout_equals_repeated_steps(monkeypatch, ns, B, T, max_steps, density):
    spec = EnvSpec(16, 16, 4, 7, max_steps=max_steps)
    e1, e2 = _pair(spec, B, seed=100 + B + ns, density=density)
    a = _acts(B, 4, T, 300)
    _force(monkeypatch, ns)
    out = e2.rollout(a)
    _force(monkeypatch, 0)
    for t in range(T):
        want = e1.step(a[t])
        for n, w in zip(("obs", "dir", "reward", "terminated", "truncated"), want):
            assert torch.equal(out[n][t], w), f"ns={ns} B={B} step {t}: {n}"
    for n in ("grid", "agents", "rng", "step_count"):
        assert torch.equal(getattr(e1, n), getattr(e2, n)), n
    e1.check_errors(); e2.check_errors()
    _force(monkeypatch, ns)
    out2 = e2.rollout(a[:3].contiguous())                   # ... and on from the written-back state
    _force(monkeypatch, 0)
    for t in range(3):
        obs, *_ = e1.step(a[t])
        assert torch.equal(out2["obs"][t], obs)
    assert torch.equal(e1.grid, e2.grid)


@pytest.mark.parametrize("ns,B", [(1, 1003), (2, 1003), (2, 4096), (9, 1003), (9, 4096 + 7)])
def test_resident_rollout_with_auto_reset(monkeypatch, ns, B):
    """Layout pool + fused auto-reset inside the resident launch, the episodes about to truncate."""
    wl = workloads.make("c4", batch=B, global_batch=65536)
    T = 24
    e1, e2 = wl.make_env(DEV, auto_reset=True), wl.make_env(DEV, auto_reset=True)
    for e in (e1, e2):
        e.step_count.fill_(wl.spec.max_steps - T // 2)
    a = _acts(B, 4, T, 900)
    _force(monkeypatch, ns)
    out = e2.rollout(a, auto_reset=True)
    _force(monkeypatch, 0)
    for t in range(T):
        want = e1.step(a[t], auto_reset=True)
        for n, w in zip(("obs", "dir", "reward", "terminated", "truncated"), want):
            assert torch.equal(out[n][t], w), f"step {t}: {n}"
        assert torch.equal(out["was_reset"][t], e1.was_reset), f"step {t}: was_reset"
    for n in ("grid", "agents", "rng", "step_count", "episode"):
        assert torch.equal(getattr(e1, n), getattr(e2, n)), n
    assert int(e2.episode.sum()) >= B


@pytest.mark.parametrize("ns,B", [(1, 1003), (2, 1003), (2, 2048 + 16)])
def test_resident_persistent_equals_repeated_steps(monkeypatch, ns, B):
    spec = EnvSpec(16, 16, 4, 7, max_steps=40)
    e_ref, e_per = _pair(spec, B, seed=7 + ns, density=0.25)
    T = 50
    a = _acts(B, 4, T, 700)
    _force(monkeypatch, ns)
    with e_per.persistent(max_steps=T + 3) as ps:
        assert ps.waves == -(-B // (16 * (2 if ns == 2 else 1)))
        for t in range(T):
            want = [x.clone() for x in e_ref.step(a[t])]
            got = ps.step(a[t])
            for n, w, g in zip(("obs", "dir", "reward", "terminated", "truncated"), want, got):
                assert torch.equal(w, g), f"ns={ns} step {t}: {n}"
    _force(monkeypatch, 0)
    assert ps.timeouts == 0 and ps.waves_left == ps.waves and ps.steps_completed == T
    for n in ("cells", "agents", "rng", "step_count"):
        assert torch.equal(getattr(e_ref, n), getattr(e_per, n)), n


def test_c4_full_size_rollout_vs_oracle():
    """BASELINE.json configs[3] at its full size through ONE launch of the resident kernel the library picks at this batch by itself
    (one slice of 16 envs per wavefront, 64 view slots: 4096 wavefronts), auto-reset fused in, against the CPU oracle: every output of
    every step, the state afterwards."""
    wl = workloads.make("c4")
    spec, B, A, T = wl.spec, wl.batch, wl.spec.num_agents, 16
    assert B == 65536
    env = wl.make_env(DEV, auto_reset=True)
    info = _lib.launch_info(spec, B, roll=True)
    assert info["resident_shape"] == 9 and info["slices"] == 1 and info["envs_per_wavefront"] == 16 and info["wavefronts"] == 4096
    assert info["lds_bytes"] == 4 * 9872                           # 16 wavefronts per CU: the whole batch is ONE resident round
    assert _lib.launch_info(spec, 49152, roll=True)["resident_shape"] == 7      # (12 wavefronts per CU while they hold the batch)
    assert _lib.launch_info(spec, 16384, roll=True)["resident_shape"] == 0
    info = _lib.launch_info(spec, B, persistent=True)            # (the closed loop must also leave VGPRs to its feeders: two slices)
    assert info["resident_shape"] == 8 and info["slices"] == 2 and info["envs_per_wavefront"] == 32 and info["wavefronts"] == 2048
    env.step_count.fill_(spec.max_steps - T // 2)
    ref = dict(grid=wl.grid.copy(), agents=wl.agents.copy(), rng=wl.rng.copy(),
               step_count=np.full(B, spec.max_steps - T // 2, np.int32), aux=None)
    episode = np.zeros(B, np.int32)
    r = np.random.default_rng(6)
    acts = r.integers(0, 7, size=(T, B, A)).astype(np.int8)
    out = env.rollout(torch.from_numpy(acts).to(DEV), auto_reset=True)
    sd, nt = spec.as_dict(), ob.max_threads()
    for t in range(T):
        was = oracle_reset_done(wl, ref, episode)
        o, d, rw, te, tr = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], acts[t], None, nthreads=nt)
        ctx = f"step {t}"
        np.testing.assert_array_equal(out["was_reset"][t].cpu().numpy(), was, err_msg=ctx)
        assert out["obs"][t].cpu().numpy().tobytes() == o.tobytes(), ctx + ": obs"
        np.testing.assert_array_equal(out["dir"][t].cpu().numpy(), d, err_msg=ctx)
        assert out["reward"][t].cpu().numpy().tobytes() == rw.tobytes(), ctx + ": reward"
        np.testing.assert_array_equal(out["terminated"][t].cpu().numpy(), te, err_msg=ctx)
        np.testing.assert_array_equal(out["truncated"][t].cpu().numpy(), tr, err_msg=ctx)
    assert env.grid.cpu().numpy().tobytes() == ref["grid"].tobytes()
    np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"])
    np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref["step_count"])
    np.testing.assert_array_equal(env.rng.cpu().numpy().view(np.uint64), ref["rng"])
    np.testing.assert_array_equal(env.episode.cpu().numpy(), episode)
    assert int(episode.sum()) >= B
    env.check_errors()


@pytest.mark.parametrize("B,waves", [(65536, 2048), (49152, 1536), (32768, 2048)], ids=["c4", "c4_49152", "c4_32768"])
def test_c4_full_size_persistent_closed_loop(B, waves):
    """... and the closed loop: mgx_step_persistent holds all 65536 envs (round 5 refused this batch); T steps driven through
    post / wait == T x mgx_step_autoreset.  The resident launch owns every CU's LDS (8 wavefronts x 20112 B of 160 KiB), so a kernel
    that needs LDS cannot run BESIDE it -- the reference steps are therefore taken before the session opens; the producer / consumer
    kernels of the hand-shake (and any policy between them) must get by on what is left: 2944 B of LDS per CU, 224 VGPRs per SIMD.
    (That is also why this launch does not take kShapes 9 as the rollout does: 16 wavefronts x 128 VGPRs leave no register at all.)"""
    wl = workloads.make("c4", batch=B)              # (32768: the one-slice shape 7 at the 8 wavefronts per CU the launch may count on)
    T = 20
    e_ref, e_per = wl.make_env(DEV, auto_reset=True), wl.make_env(DEV, auto_reset=True)
    for e in (e_ref, e_per):
        e.step_count.fill_(wl.spec.max_steps - T // 2)
    a = _acts(B, 4, T, 1234)
    want = []
    for t in range(T):
        want.append([x.clone() for x in e_ref.step(a[t], auto_reset=True)] + [e_ref.was_reset.clone()])
    torch.cuda.synchronize()
    # (copies run beside the launch; a comparison is a REDUCTION, which wants LDS and would wait for the resident launch to end, and a
    # fresh allocation may wait for the device: the buffers are made before the session opens, the outputs compared after it closes)
    got = [[torch.empty_like(w) for w in want[t]] for t in range(T)]
    torch.cuda.synchronize()
    with e_per.persistent(max_steps=T, auto_reset=True) as ps:
        assert ps.waves == waves
        for t in range(T):
            for dst, src in zip(got[t], list(ps.step(a[t])) + [e_per.was_reset]):
                dst.copy_(src)
    assert ps.timeouts == 0 and ps.steps_completed == T
    for t in range(T):
        for n, w, g in zip(("obs", "dir", "reward", "terminated", "truncated", "was_reset"), want[t], got[t]):
            assert torch.equal(w, g), f"step {t}: {n}"
    for n in ("cells", "agents", "rng", "step_count", "episode"):
        assert torch.equal(getattr(e_ref, n), getattr(e_per, n)), n
