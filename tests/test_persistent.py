"""Persistent stepping (include/mgx.h: mgx_step_persistent): ONE resident launch that takes each step's actions as tagged
granules and publishes a per-wavefront flag behind its outputs == T x mgx_step, bit for bit -- every output of every step and the
state written back at the end.  Reference semantics: multigrid/base.py:303-346 called in a closed loop
(multigrid/rllib/__init__.py:59-63)."""
import zlib

import numpy as np
import pytest
import torch

import multigrid_amd as mg
from multigrid_amd import workloads
from multigrid_amd.batched import BatchedMultiGridEnv
from multigrid_amd.spec import EnvSpec

from . import util

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def _pair(spec, B, seed, density=0.25):
    st = util.random_state(spec, B, seed=seed, density=density)
    envs = []
    for _ in range(2):
        e = BatchedMultiGridEnv(spec, B, dev())
        e.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
        envs.append(e)
    return envs


def _compare_steps(e_ref, e_per, acts, ps, ctx, auto_reset=False):
    for t in range(acts.shape[0]):
        want = [x.clone() for x in e_ref.step(acts[t], auto_reset=auto_reset)]
        got = ps.step(acts[t])
        for n, w, g in zip(("obs", "dir", "reward", "terminated", "truncated"), want, got):
            assert torch.equal(w, g), f"{ctx} step {t}: {n}"
        if auto_reset:
            assert torch.equal(e_ref.was_reset, e_per.was_reset), f"{ctx} step {t}: was_reset"


SHAPES = [
    # name, spec kwargs, batch, steps
    ("c2_shape", dict(width=16, height=16, num_agents=4, view_size=7, max_steps=1024), 4096, 24),
    ("c2_ragged", dict(width=16, height=16, num_agents=4, view_size=7, max_steps=40), 1003, 60),
    ("share8", dict(width=16, height=16, num_agents=4, view_size=7, max_steps=1024), 8192, 12),
    ("a1_v3", dict(width=9, height=7, num_agents=1, view_size=3, max_steps=30), 300, 40),
    ("a3_v5_nooverlap", dict(width=8, height=8, num_agents=3, view_size=5, max_steps=50, allow_agent_overlap=False), 500, 30),
    ("a5_v9", dict(width=20, height=12, num_agents=5, view_size=9, max_steps=64, success_termination_mode="all", failure_termination_mode="any"), 257, 30),
    ("a16_v9", dict(width=32, height=32, num_agents=16, view_size=9, max_steps=64, joint_reward=True), 96, 16),
    ("a2_v15_seethrough", dict(width=24, height=24, num_agents=2, view_size=15, max_steps=64, see_through_walls=True), 200, 12),
]


@pytest.mark.parametrize("name,kw,B,T", SHAPES, ids=[s[0] for s in SHAPES])
def test_persistent_equals_repeated_steps(name, kw, B, T):
    spec = EnvSpec(**kw)
    e_ref, e_per = _pair(spec, B, seed=zlib.crc32(name.encode()) % 1000)
    acts = torch.from_numpy(np.stack([util.random_actions(B, spec.num_agents, seed=700 + t) for t in range(T)])).to(dev())
    with e_per.persistent(max_steps=T + 5) as ps:             # (closed early: a stop request ends the launch)
        _compare_steps(e_ref, e_per, acts, ps, name)
    assert ps.timeouts == 0 and ps.waves_left == ps.waves and ps.steps_completed == T
    for n in ("cells", "agents", "rng", "step_count"):
        assert torch.equal(getattr(e_ref, n), getattr(e_per, n)), f"{name}: {n} after the session"
    e_ref.check_errors(); e_per.check_errors()
    # ... and the env steps on normally from the written-back state
    a = acts[0]
    for x, y in zip(e_ref.step(a), e_per.step(a)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("wl_name,B,T", [("c2", 4096, 40), ("c3", 16384, 40)])
def test_persistent_with_auto_reset_on_the_bench_workloads(wl_name, B, T):
    """The bench's workloads (layout pool, fused auto-reset; C3: BlockedUnlockPickup's hook) through the persistent launch, all of
    max_steps used (the launch ends by itself)."""
    wl = workloads.make(wl_name, batch=B, global_batch=B)
    e_ref, e_per = wl.make_env(dev(), auto_reset=True), wl.make_env(dev(), auto_reset=True)
    # (close to the end of the episodes, so that truncation resets happen inside the session)
    for e in (e_ref, e_per):
        e.step_count.fill_(wl.spec.max_steps - T // 2)
    acts = torch.from_numpy(np.stack([util.random_actions(B, wl.spec.num_agents, seed=900 + t, p_missing=0.0) for t in range(T)])).to(dev())
    with e_per.persistent(max_steps=T, auto_reset=True) as ps:
        _compare_steps(e_ref, e_per, acts, ps, wl_name, auto_reset=True)
    assert ps.steps_completed == T
    for n in ("cells", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(e_ref, n), getattr(e_per, n)), f"{wl_name}: {n} after the session"
    assert int(e_per.episode.sum()) >= B          # every env went through its truncation reset


def test_feed_plays_a_recorded_sequence_in_closed_loop():
    """mgx_persistent_feed (one resident workgroup: wait for step t-1, post step t) drives the whole session; the final outputs
    and state equal T x mgx_step; the trace is monotonic."""
    spec = EnvSpec(width=16, height=16, num_agents=4, view_size=7, max_steps=1024)
    B, T = 4096, 200
    e_ref, e_per = _pair(spec, B, seed=11)
    acts = torch.from_numpy(np.stack([util.random_actions(B, 4, seed=40 + t) for t in range(T)])).to(dev())
    for t in range(T):
        want = e_ref.step(acts[t])
    with e_per.persistent(max_steps=T) as ps:
        tr = ps.feed(acts, trace=True)
    assert ps.timeouts == 0 and ps.steps_completed == T
    for n, w in zip(("obs", "dir", "reward", "terminated", "truncated"), want):
        assert torch.equal(w, getattr(e_per, n)), n
    for n in ("cells", "agents", "rng", "step_count"):
        assert torch.equal(getattr(e_ref, n), getattr(e_per, n)), n
    tr = tr.cpu().numpy()
    assert (np.diff(tr) >= 0).all() and tr[0] > 0
    per_step_us = (tr[-1] - tr[0]) / T / 100.0
    assert 1.0 < per_step_us < 200.0, per_step_us


def test_persistent_runs_beside_its_producer_whatever_streams_exist():
    """The persistent launch and its producer must never serialise (HIP maps streams onto a few hardware queues; the launch has a
    high-priority stream of its own).  Sessions opened after many other streams were created; the actions are posted from a side
    stream, the outputs awaited on the default stream."""
    spec = EnvSpec(width=16, height=16, num_agents=4, view_size=7, max_steps=1024)
    B, T = 2048, 6
    keep = []
    main = torch.cuda.current_stream(dev())
    for n_streams in (1, 2, 3, 5, 8):
        keep += [torch.cuda.Stream(dev()) for _ in range(n_streams)]
        for s in keep[-2:]:
            with torch.cuda.stream(s):
                torch.zeros(16, device=dev()).add_(1)
        e_ref, e_per = _pair(spec, B, seed=n_streams)
        acts = torch.from_numpy(np.stack([util.random_actions(B, 4, seed=t) for t in range(T)])).to(dev())
        side = keep[-1]
        torch.cuda.synchronize()
        with e_per.persistent(max_steps=T, timeout_ms=500) as ps:
            for t in range(T):
                want = [x.clone() for x in e_ref.step(acts[t])]
                side.wait_stream(main)                      # (the producer's stream has seen the previous step's outputs)
                with torch.cuda.stream(side):
                    ps.post(acts[t])
                got = ps.wait()
                for w, g in zip(want, got):
                    assert torch.equal(w, g), (n_streams, t)
        assert ps.timeouts == 0


def test_a_producer_that_goes_away_does_not_hang_the_device():
    """No actions are ever posted: every wavefront gives up after timeout_ms, writes its (unchanged) state back and leaves."""
    spec = EnvSpec(width=16, height=16, num_agents=4, view_size=7, max_steps=1024)
    e_ref, e_per = _pair(spec, 2048, seed=3)
    ps = e_per.persistent(max_steps=10, timeout_ms=100)
    ps.__enter__()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="timed out"):
        ps.t = ps.waited = ps.max_steps            # (nothing to stop: the launch has ended by timeout)
        ps.close()
    assert ps.timeouts == ps.waves and ps.steps_completed == 0
    for n in ("cells", "agents", "rng", "step_count"):
        assert torch.equal(getattr(e_ref, n), getattr(e_per, n)), n
    a = torch.from_numpy(util.random_actions(2048, 4, seed=1)).to(dev())
    for x, y in zip(e_ref.step(a), e_per.step(a)):
        assert torch.equal(x, y)


def test_persistent_refuses_what_it_cannot_hold():
    from multigrid_amd import _lib
    # 131072 envs of the C4 shape: more wavefronts than the chip holds at once (round 6: its 65536 envs ARE held -- two slices of 16
    # envs per wavefront, tests/test_resident.py)
    wl = workloads.make("c4", batch=131072, global_batch=131072)
    env = wl.make_env(dev(), auto_reset=True)
    with pytest.raises(_lib.MgxError) as ei:
        env.persistent(max_steps=4, auto_reset=True)
    assert ei.value.code == _lib.ERR_UNSUPPORTED
    env2 = workloads.make("c2").make_env(dev(), auto_reset=True)
    with env2.persistent(max_steps=4, auto_reset=True):
        with pytest.raises(RuntimeError, match="persistent session is open"):
            env2.step(torch.zeros((4096, 4), dtype=torch.int8, device=dev()))
        # ... and so is everything that reads or replaces the state tensors: they are stale until the launch writes them back
        for call in (env2.state_dict, lambda: env2.seed(3), lambda: env2.seed_synthetic(3), env2.is_done, env2.gen_obs,
                     lambda: env2.load_state_dict({}), lambda: env2.set_layout_pool(None, None),
                     lambda: env2.load_state(None, None)):
            with pytest.raises(RuntimeError, match="persistent session is open"):
                call()
    env2.state_dict(); env2.is_done()                  # (closed: available again)
