"""BASELINE.json's GPU configurations at their FULL sizes, HIP vs the CPU oracle, every step, every output and the whole
post-step state (multigrid/base.py:303-346, utils/obs.py:65-102):

    C4  Empty-16x16, 4 agents, 65536 envs                                  24 steps
    C3  BlockedUnlockPickup, 2 agents, 16384 envs on the real K=256 layout pool, half of the envs one pickup away from
        the target box so that the run goes THROUGH success events and the fused auto-reset     32 steps
    C5  64x64 with occluders, 16 agents, view 9, 32768 envs                  10 steps
    C2  Empty-16x16, 4 agents, 4096 envs, with auto-reset, long enough that goals are reached     64 steps

The HIP side runs the fused auto-reset step (what bench.py times); the oracle side emulates the restart from its
definition (include/mgx.h: layout = (first_env + b + episode * 7919) mod K) in numpy and then takes the oracle step.
The workloads are the ones bench.py times (multigrid_amd/workloads.py)."""
import numpy as np
import pytest
import torch

from multigrid_amd import workloads
from oracle import binding as ob

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def oracle_reset_done(wl, ref, episode):
    """mgx_reset_done's definition (include/mgx.h) on the oracle's numpy state.  Returns was_reset u8[B]."""
    spec = wl.spec
    pg, pa, pt = wl.pool
    K = pg.shape[0]
    done = (ref["agents"][:, :, 4] != 0).all(axis=1) | (ref["step_count"] >= spec.max_steps)
    idx = np.nonzero(done)[0]
    if len(idx):
        k = (wl.first_env + idx + episode[idx].astype(np.int64) * 7919) % K
        ref["grid"][idx] = pg[k]
        ref["agents"][idx] = pa[k]
        if pt is not None:
            ref["aux"][idx] = pt[k]
        ref["step_count"][idx] = 0
        episode[idx] += 1
    return done.astype(np.uint8)


def run_vs_oracle(wl, T, seed, action_p=None, min_resets=0, min_rewards=0):
    spec, B, A = wl.spec, wl.batch, wl.spec.num_agents
    env = wl.make_env(DEV, auto_reset=True)
    ref = dict(grid=wl.grid.copy(), agents=wl.agents.copy(), rng=wl.rng.copy(), step_count=np.zeros(B, np.int32),
               aux=None if wl.aux is None else wl.aux.copy())
    episode = np.zeros(B, np.int32)
    sd = spec.as_dict()
    nt = ob.max_threads()
    r = np.random.default_rng(seed)
    n_resets = n_rewards = 0
    for t in range(T):
        act = r.choice(7, size=(B, A), p=action_p).astype(np.int8)
        was_ref = oracle_reset_done(wl, ref, episode)
        o_ref, d_ref, r_ref, te_ref, tr_ref = ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"],
                                                            act, ref["aux"], nthreads=nt)
        obs, dirs, rew, term, trunc = env.step(torch.from_numpy(act).to(DEV), auto_reset=True)
        ctx = f"{wl.name} step {t}"
        np.testing.assert_array_equal(env.was_reset.cpu().numpy(), was_ref, err_msg=ctx)
        assert obs.cpu().numpy().tobytes() == o_ref.tobytes(), ctx + ": obs"
        np.testing.assert_array_equal(dirs.cpu().numpy(), d_ref, err_msg=ctx)
        assert rew.cpu().numpy().tobytes() == r_ref.tobytes(), ctx + ": reward"
        np.testing.assert_array_equal(term.cpu().numpy(), te_ref, err_msg=ctx)
        np.testing.assert_array_equal(trunc.cpu().numpy(), tr_ref, err_msg=ctx)
        assert env.grid.cpu().numpy().tobytes() == ref["grid"].tobytes(), ctx + ": grid"
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"], err_msg=ctx)
        np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref["step_count"], err_msg=ctx)
        np.testing.assert_array_equal(env.rng.cpu().numpy().view(np.uint64), ref["rng"], err_msg=ctx)
        if ref["aux"] is not None:
            np.testing.assert_array_equal(env.aux.cpu().numpy(), ref["aux"], err_msg=ctx)
        n_resets += int(was_ref.sum())
        n_rewards += int((r_ref > 0).sum())
    np.testing.assert_array_equal(env.episode.cpu().numpy(), episode)
    env.check_errors()
    assert n_resets >= min_resets, f"only {n_resets} restarts happened"
    assert n_rewards >= min_rewards, f"only {n_rewards} rewards were paid"
    return n_resets, n_rewards


def test_c4_full_size_vs_oracle():
    wl = workloads.make("c4")
    assert wl.batch == 65536
    run_vs_oracle(wl, T=24, seed=4)


@pytest.mark.parametrize("form", ["graph", "eager_chains"])
def test_c4_full_size_four_chains_vs_oracle(form):
    """The sub-sharded forms at full size (bench.py's `pipelined` point is the policy's two chains; four are the finer cut, and
    the unjoined eager steps below run the policy's own): C4 at 65 536 envs stepped as 4 independent chains of 16 384-env
    launches (capture_steps(sub_shards=4): four parallel branches of one hipGraph; or step(..., sub_shards=4): mgx_step_chains on
    four streams), auto-reset fused in -- against the ORACLE, every output and the whole state:
      * a one-step 4-chain graph replayed per step with fresh actions: every step's outputs and post-step state;
      * then a 12-step 4-chain graph replayed once (the chains free to drift apart, as in the bench): the last step's outputs
        and the whole state after the run."""
    wl = workloads.make("c4")
    assert wl.batch == 65536
    spec, B, A = wl.spec, wl.batch, wl.spec.num_agents
    env = wl.make_env(DEV, auto_reset=True)
    assert env.sub_shards_hint(auto_reset=True) == 2                     # (what capture_steps(sub_shards="auto") resolves to on an MI355X: round 5)
    assert env.sub_shards_hint(auto_reset=True, form="eager") == 1       # (round 6: per-step calls from Python are host-bound as chains)
    ref = dict(grid=wl.grid.copy(), agents=wl.agents.copy(), rng=wl.rng.copy(), step_count=np.zeros(B, np.int32), aux=None)
    episode = np.zeros(B, np.int32)
    sd, nt = spec.as_dict(), ob.max_threads()
    r = np.random.default_rng(44)

    def oracle_step(act):
        was = oracle_reset_done(wl, ref, episode)
        return was, ob.step_batch(sd, ref["grid"], ref["agents"], ref["rng"], ref["step_count"], act, None, nthreads=nt)

    def check(ctx, was_ref, outs_ref):
        torch.cuda.synchronize()
        o_ref, d_ref, r_ref, te_ref, tr_ref = outs_ref
        np.testing.assert_array_equal(env.was_reset.cpu().numpy(), was_ref, err_msg=ctx)
        assert env.obs.cpu().numpy().tobytes() == o_ref.tobytes(), ctx + ": obs"
        np.testing.assert_array_equal(env.dir.cpu().numpy(), d_ref, err_msg=ctx)
        assert env.reward.cpu().numpy().tobytes() == r_ref.tobytes(), ctx + ": reward"
        np.testing.assert_array_equal(env.terminated.cpu().numpy(), te_ref, err_msg=ctx)
        np.testing.assert_array_equal(env.truncated.cpu().numpy(), tr_ref, err_msg=ctx)
        assert env.grid.cpu().numpy().tobytes() == ref["grid"].tobytes(), ctx + ": grid"
        np.testing.assert_array_equal(env.agents.cpu().numpy(), ref["agents"], err_msg=ctx)
        np.testing.assert_array_equal(env.step_count.cpu().numpy(), ref["step_count"], err_msg=ctx)
        np.testing.assert_array_equal(env.rng.cpu().numpy().view(np.uint64), ref["rng"], err_msg=ctx)
        np.testing.assert_array_equal(env.episode.cpu().numpy(), episode, err_msg=ctx)

    T1, T2 = 6, 12
    if form == "graph":
        buf = torch.zeros((1, B, A), dtype=torch.int8, device=DEV)
        g1 = env.capture_steps(buf, auto_reset=True, sub_shards=4)
        assert g1.sub_shards == 4
        for t in range(T1):
            act = r.integers(0, 7, size=(B, A)).astype(np.int8)
            was, outs = oracle_step(act)
            buf[0].copy_(torch.from_numpy(act).to(DEV))
            g1.replay()
            check(f"one-step 4-chain graph, step {t}", was, outs)
        acts = r.integers(0, 7, size=(T2, B, A)).astype(np.int8)
        g12 = env.capture_steps(torch.from_numpy(acts).to(DEV), auto_reset=True, sub_shards=4)
        for t in range(T2):
            was, outs = oracle_step(acts[t])
        g12.replay()
        check("12-step 4-chain graph", was, outs)
    else:
        acts = r.integers(0, 7, size=(T1 + T2, B, A)).astype(np.int8)
        dacts = torch.from_numpy(acts).to(DEV)
        for t in range(T1):                                              # joined after every step
            was, outs = oracle_step(acts[t])
            env.step(dacts[t], auto_reset=True, sub_shards=4)
            env.join()
            check(f"eager 4 chains, step {t}", was, outs)
        for t in range(T1, T1 + T2):                                     # free-running chains, joined once
            was, outs = oracle_step(acts[t])
            env.step(dacts[t], auto_reset=True, sub_shards=2)            # (the graph policy's two chains, issued eagerly)
        env.join()
        check("eager 2 chains, 12 steps unjoined", was, outs)
    env.check_errors()


def test_c2_with_auto_reset_through_goals_vs_oracle():
    """Agents biased to walk (forward 55 %): goals are reached, envs restart, the run continues -- all vs the oracle."""
    wl = workloads.make("c2")
    # start the agents near the goal corner so that 64 steps are enough for many successes
    wl.agents[:, :, 2] = 11 + (np.arange(wl.batch)[:, None] + np.arange(4)[None]) % 3
    wl.agents[:, :, 3] = 12 + (np.arange(wl.batch)[:, None] // 3 + np.arange(4)[None]) % 3
    p = np.array([0.1, 0.1, 0.55, 0.05, 0.05, 0.05, 0.1])
    run_vs_oracle(wl, T=64, seed=2, action_p=p, min_resets=1000, min_rewards=1000)


def test_c3_full_size_layout_pool_through_success_vs_oracle():
    wl = workloads.make("c3")
    assert wl.batch == 16384 and wl.pool[0].shape[0] == 256
    # every other env: agent 0 stands next to the target box, facing it, hands empty -> `pickup` ends the episode with a
    # joint reward through the env hook (blockedunlockpickup.py:166-175), then the env restarts from the pool
    g = wl.grid
    moved = 0
    for b in range(0, wl.batch, 2):
        (by, bx), = np.argwhere(g[b, :, :, 0] == 7)
        for d, (dx, dy) in enumerate(((1, 0), (0, 1), (-1, 0), (0, -1))):           # agent at box - d, facing d
            ax, ay = bx - dx, by - dy
            if g[b, ay, ax, 0] == 1:
                wl.agents[b, 0, 1:4] = (d, ax, ay)
                moved += 1
                break
    assert moved > wl.batch // 4
    n_resets, n_rewards = run_vs_oracle(wl, T=32, seed=3, min_resets=2000, min_rewards=4000)


def test_c5_full_size_with_occluders_vs_oracle():
    wl = workloads.make("c5")
    assert wl.batch == 32768
    t = wl.grid[..., 0]
    assert (t[:, 1:-1, 1:-1] == 2).mean() > 0.04 and (t == 4).any() and (t == 5).any()       # the occluders are there
    run_vs_oracle(wl, T=10, seed=5)
