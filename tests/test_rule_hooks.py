"""The DECLARED step() hook (include/mgx.h: MGX_KIND_RULES, EnvSpec(env_kind="rules")): a user-defined env's post-step hook --
`if agent.state.carrying == self.obj: on_success`, `if action == toggle and fwd_obj == self.door and self.door.is_open: on_failure`
(the two shapes the reference's own hooks have: envs/blockedunlockpickup.py:166-175, envs/redbluedoors.py:170-187) -- as a small
table in `aux` that the fused kernel evaluates, so that such an env steps on the BATCHED engine too.

  * BlockedUnlockPickup re-expressed as the one-rule table {carries (box, colour) -> success} replays the reference's own `bup_*`
    fixtures bit for bit (oracle, host rules, and the HIP kernels);
  * the FetchTrapEnv of tests/custom_envs.py -- whose `step` override was recorded over the REAL reference -- runs on the batched
    engine from its table {carries (ball, purple) -> success; toggles_at (trap door) while open -> failure} and reproduces every
    recorded reward, termination, observation and state."""
import dataclasses
import os

import numpy as np
import pytest
import torch

from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts
from oracle import binding as ob
from tests import custom_envs, hostshim, util

BUP = [p for p in util.GOLDEN if os.path.basename(p).startswith("bup_")]
FETCH = [p for p in util.CUSTOM_STEPS_GOLDEN if "fetchtrap" in p]


def bup_as_rules(path):
    z, d, spec = util.load_golden(path)
    assert spec.env_kind == "blockedunlockpickup"
    aux = layouts.rules_aux([("carries", d["target"][0], d["target"][1], "success")])
    return z, dataclasses.replace(spec, env_kind="rules"), aux


def test_the_bup_fixtures_do_reach_the_hook():
    assert sum(bool((np.load(p)["reward"] > 0).any()) for p in BUP) >= 1


def test_rules_aux_layout():
    aux = layouts.rules_aux([("carries", 6, 3, "success"), ("toggles_at", 5, 4, "failure", "open")])
    assert aux.tolist() == [2, 1, 6, 3, 1, 0, 2, 5, 4, 2, 1, 0, 0, 0, 0, 0]
    with pytest.raises(ValueError):
        layouts.rules_aux([("carries", 1, 1, "success")] * 4)


@pytest.mark.parametrize("path", BUP, ids=[os.path.basename(p)[:-4] for p in BUP])
def test_blockedunlockpickup_as_one_rule_replays_the_reference_on_oracle_and_host_rules(path):
    assert BUP
    z, spec, aux = bup_as_rules(path)
    # the oracle's rule kind against the reference's fixture ...
    env = ob.RefEnv(spec.as_dict(), z["grid0"], z["agents0"], util.rng_words_lohi(z["rng0"])[[1, 0, 3, 2]], target=[int(v) for v in aux])
    # ... and the host build of the kernels' rules
    tile = layouts.grid_to_product(z["grid0"]); rows = layouts.pack_agents(z["agents0"])
    rng = util.rng_words_lohi(z["rng0"]); haux = aux.copy(); sc = 0
    for t in range(z["actions"].shape[0]):
        o, d_, r, te, tr, _ = env.step(np.ascontiguousarray(z["actions"][t]))
        out = hostshim.step_env(spec, tile, rows, np.ascontiguousarray(z["actions"][t]), rng, sc, haux)
        sc = out["step_count"]
        ctx = f"step {t}"
        for got_obs, got_r, got_te in ((o, r, te), (out["obs"], out["reward"], out["terminated"])):
            np.testing.assert_array_equal(got_obs, z["obs"][t], err_msg=ctx)
            assert np.asarray(got_r, dtype=np.float64).tobytes() == z["reward"][t].tobytes(), ctx
            np.testing.assert_array_equal(np.asarray(got_te).astype(np.uint8), z["terminated"][t], err_msg=ctx)
        np.testing.assert_array_equal(layouts.grid_from_product(tile), z["grid"][t].astype(np.int64), err_msg=ctx)
        np.testing.assert_array_equal(layouts.unpack_agents(rows), z["agents"][t].astype(np.int64), err_msg=ctx)


def _fetchtrap_batched(path, device, backend=None):
    z = np.load(path)
    cname, kw, T = custom_envs.ALL_STEP_CASES[os.path.basename(path)[:-4]]
    import json
    d = json.loads(str(z["spec_json"]))
    spec = dataclasses.replace(EnvSpec.from_dict(d), env_kind="rules")
    A = spec.num_agents
    n_succ = n_fail = 0
    for k in range(len(z["reset_seeds"])):
        g0 = z["grid0"][k].astype(np.int64)
        (tx, ty), = np.argwhere((g0[..., 0] == 4) & (g0[..., 1] == 0))                 # the red trap door
        aux = layouts.rules_aux([("carries", 6, 3, "success"), ("toggles_at", int(tx), int(ty), "failure", "open")])
        env = BatchedMultiGridEnv(spec, 2, device, backend=backend(spec) if backend else None)
        env.load_state(np.stack([layouts.grid_to_product(g0)] * 2), np.stack([layouts.pack_agents(z["agents0"][k])] * 2),
                       rng=np.stack([util.rng_words_lohi(z["rng0"][k])] * 2), aux=np.stack([aux] * 2))
        obs, _ = env.gen_obs()
        np.testing.assert_array_equal(obs[1].cpu().numpy(), z["obs0"][k])
        for t in range(T):
            act = torch.from_numpy(np.stack([z["actions"][k][t]] * 2)).to(device)
            obs, dirs, rew, term, trunc = env.step(act)
            ctx = f"episode {k} step {t}"
            np.testing.assert_array_equal(obs[1].cpu().numpy(), z["obs"][k][t], err_msg=ctx)
            assert rew[0].cpu().numpy().tobytes() == z["reward"][k][t].tobytes(), (ctx, rew[0], z["reward"][k][t])
            np.testing.assert_array_equal(term[1].cpu().numpy(), z["terminated"][k][t], err_msg=ctx)
            assert int(trunc[0]) == int(z["truncated"][k][t]), ctx
            np.testing.assert_array_equal(layouts.grid_from_product(env.grid[0].cpu().numpy()), z["grid"][k][t].astype(np.int64), err_msg=ctx)
            np.testing.assert_array_equal(layouts.unpack_agents(env.agents[1].cpu().numpy()), z["agents"][k][t].astype(np.int64), err_msg=ctx)
            n_succ += bool((z["reward"][k][t] > 0).any()); n_fail += bool(z["terminated"][k][t].any() and not (z["reward"][k][t] > 0).any())
        env.check_errors()
    assert n_succ > 10 and n_fail > 3, (n_succ, n_fail)


@pytest.mark.parametrize("path", FETCH, ids=[os.path.basename(p)[:-4] for p in FETCH])
def test_user_step_hook_as_rules_on_the_batched_engine_matches_the_reference_cpu(path):
    assert FETCH
    _fetchtrap_batched(path, "cpu", backend=lambda spec: util.OracleBackend(spec))


@pytest.mark.gpu
@pytest.mark.parametrize("path", FETCH, ids=[os.path.basename(p)[:-4] for p in FETCH])
def test_user_step_hook_as_rules_on_the_batched_engine_matches_the_reference_gpu(path):
    _fetchtrap_batched(path, "cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("path", BUP, ids=[os.path.basename(p)[:-4] for p in BUP])
def test_blockedunlockpickup_as_one_rule_replays_the_reference_on_gpu(path):
    z, spec, aux = bup_as_rules(path)
    B = 3
    env = BatchedMultiGridEnv(spec, B, "cuda:0")
    env.load_state(np.stack([layouts.grid_to_product(z["grid0"])] * B), np.stack([layouts.pack_agents(z["agents0"])] * B),
                   rng=np.stack([util.rng_words_lohi(z["rng0"])] * B), aux=np.stack([aux] * B))
    for t in range(z["actions"].shape[0]):
        obs, dirs, rew, term, trunc = env.step(torch.from_numpy(np.stack([z["actions"][t]] * B)).to("cuda:0"))
        ctx = f"step {t}"
        np.testing.assert_array_equal(obs[B - 1].cpu().numpy(), z["obs"][t], err_msg=ctx)
        assert rew[0].cpu().numpy().tobytes() == z["reward"][t].tobytes(), ctx
        np.testing.assert_array_equal(term[1].cpu().numpy(), z["terminated"][t], err_msg=ctx)
        np.testing.assert_array_equal(layouts.unpack_agents(env.agents[2].cpu().numpy()), z["agents"][t].astype(np.int64), err_msg=ctx)
    env.check_errors()


@pytest.mark.gpu
def test_rules_kind_equals_compiled_kind_on_random_blockedunlockpickup_states():
    """Large batch, throughput instantiation, fused auto-reset: env_kind 'rules' with the one-rule table == env_kind
    'blockedunlockpickup' with its target box, on every output and the whole state."""
    from multigrid_amd import workloads
    wl = workloads.make("c3")
    spec_r = dataclasses.replace(wl.spec, env_kind="rules")
    aux_r = np.stack([layouts.rules_aux([("carries", a[0], a[1], "success")]) for a in wl.aux])
    pool_aux_r = np.stack([layouts.rules_aux([("carries", a[0], a[1], "success")]) for a in wl.pool[2]])
    a = wl.make_env("cuda:0", auto_reset=True)
    b = BatchedMultiGridEnv(spec_r, wl.batch, "cuda:0")
    b.load_state(wl.grid, wl.agents, rng=wl.rng, aux=aux_r, validate=False)
    b.set_layout_pool(wl.pool[0], wl.pool[1], pool_aux_r)
    for env in (a, b):
        env.step_count.fill_(wl.spec.max_steps - 6)
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    for t in range(12):
        act = torch.randint(0, 7, (wl.batch, 2), dtype=torch.int8, device="cuda:0", generator=g)
        x, y = a.step(act, auto_reset=True), b.step(act, auto_reset=True)
        for u, v in zip(x, y):
            assert torch.equal(u, v), f"step {t}"
    for f in ("cells", "agents", "rng", "step_count", "episode"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert int(a.episode.sum()) > 1000
