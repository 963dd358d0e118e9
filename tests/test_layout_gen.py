"""On-device episode starts (mgx_reset_generate).

CPU (`not gpu`): the chain that pins the generator to the reference --
  numpy itself  ==  the oracle's restatement of Generator(PCG64).integers (mixed with random() calls, buffered halves)
  layouts.py (pinned by the reference's reset fixtures, tests/test_env_compat.py)  ==  oracle/mgx_layout_oracle.c
GPU: the HIP kernel == the oracle, env by env, over many episodes, interleaved with steps; and the generated episodes
have the reference's structure (box right, locked door + matching key, ball in front of the door, agents left)."""
import numpy as np
import pytest
import torch

from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts, rng as rnglib
from oracle import binding as ob
from tests import util


def test_oracle_integers_equal_numpy():
    r = np.random.default_rng(123)
    for trial in range(200):
        g = np.random.Generator(np.random.PCG64(int(r.integers(0, 2 ** 40))))
        for k in range(int(r.integers(0, 4))):
            g.random()
        if r.random() < 0.5:
            g.integers(0, 5)                                   # leaves a buffered 32-bit half behind
        w = ob.gen_words(g)
        np.testing.assert_array_equal(w, rnglib.gen_words_from_generator(g))
        for k in range(20):
            lo = int(r.integers(-5, 5))
            hi = lo + (int(r.integers(1, 300)) if r.random() < 0.9 else int(r.integers(1, 2 ** 31)))
            assert int(g.integers(lo, hi)) == int(ob.np_integers(w, lo, hi, 1)[0])
            if r.random() < 0.3:                               # random() draws 64 bits and leaves the buffer alone
                g.random(); tmp = w[:4].copy(); ob.pcg64_random(tmp, 1); w[:4] = tmp
        np.testing.assert_array_equal(ob.gen_words(g), w)
        g2 = rnglib.generator_from_gen_words(w)
        assert int(g2.integers(0, 1000)) == int(g.integers(0, 1000))


def test_group_jump_constants_and_the_skipped_generator_equal_numpy():
    """mgx_layout_gen.h `group_init`'s table (lane j of a candidate's group evaluates try j of a place_obj call): row j must be
    {mult^j, 1 + mult + ... + mult^(j-1)} mod 2^128 -- and a generator advanced that way must stand where numpy's stands after j tries
    of two bounded 32-bit draws each, whichever half of a word was pending (has_uint32 kept, uinteger = the high half of the last word)."""
    import os
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multigrid_amd", "csrc", "mgx_layout_gen.h")).read()
    body = src[src.index("static constexpr uint64_t T[kGroupLanes][4]"):]
    body = body[:body.index("};")]
    rows = [[int(x, 16) for x in re.findall(r"0x([0-9A-Fa-f]{16})ull", line)] for line in body.splitlines() if "0x" in line]
    assert len(rows) == 8 and all(len(r) == 4 for r in rows)
    mult, mask = 0x2360ED051FC65DA44385DF649FCCF645, (1 << 128) - 1
    a, c = 1, 0
    for j in range(8):
        assert rows[j][0] | (rows[j][1] << 64) == a and rows[j][2] | (rows[j][3] << 64) == c, j
        a, c = (a * mult) & mask, (c * mult + 1) & mask
    r = np.random.default_rng(5)
    for trial in range(50):
        g = np.random.Generator(np.random.PCG64(int(r.integers(0, 2 ** 40))))
        if trial & 1:
            g.integers(0, 5)                                   # a pending high half
        st = g.bit_generator.state
        state, inc, has, uint = st["state"]["state"], st["state"]["inc"], st["has_uint32"], st["uinteger"]
        for j in range(8):
            if j:                                              # group_skip: j words on, the pending flag kept
                aj, cj = rows[j][0] | (rows[j][1] << 64), rows[j][2] | (rows[j][3] << 64)
                s2 = (state * aj + cj * inc) & mask
                hi, lo = s2 >> 64, s2 & ((1 << 64) - 1)
                rot = hi >> 58
                x = hi ^ lo
                out = ((x >> rot) | (x << ((64 - rot) & 63))) & ((1 << 64) - 1)
                want = dict(state=s2, has=has, uint=out >> 32)
            else:
                want = dict(state=state, has=has, uint=uint)
            st2 = g.bit_generator.state
            assert st2["state"]["state"] == want["state"] and st2["has_uint32"] == want["has"], (trial, j)
            if j or has:
                assert st2["uinteger"] == want["uint"], (trial, j)
            g.integers(0, 7); g.integers(0, 11)                # one try: two bounded draws (a re-sample is ~1e-9: not here)


def test_layout_oracle_equals_layouts_py():
    r = np.random.default_rng(7)
    blank = layouts.roomgrid_blank(6, 1, 2)
    for seed in range(300):
        A = int(r.integers(1, 5))
        lg, ng = np.random.default_rng(1000 + seed), np.random.default_rng(2000 + seed)
        if seed % 3 == 0:
            ng.integers(0, 7)
        lw, nw = ob.gen_words(lg), ob.gen_words(ng)
        g_ref, a_ref, t_ref = layouts.blockedunlockpickup_layout(6, A, lg, ng)
        g, a, aux = ob.bup_layout(6, A, lw, nw, blank)
        np.testing.assert_array_equal(g, g_ref); np.testing.assert_array_equal(a, a_ref)
        np.testing.assert_array_equal(aux[:3], t_ref[:3])
        np.testing.assert_array_equal(lw, ob.gen_words(lg)); np.testing.assert_array_equal(nw, ob.gen_words(ng))
    for seed in range(200):                                                    # RedBlueDoors (redbluedoors.py:142-168)
        size = int(r.integers(4, 11))
        A = int(r.integers(1, min(6, (size - 2) ** 2)))                      # (the room's interior must hold the agents)
        lg = np.random.default_rng(7000 + seed)
        if seed % 4 == 0:
            lg.integers(0, 9)                                                  # a buffered 32-bit half
        lw = ob.gen_words(lg)
        g_ref, a_ref = layouts.redbluedoors_layout(size, A, lg)
        g, a, aux = ob.rbd_layout(size, A, lw, layouts.redbluedoors_blank(size))
        np.testing.assert_array_equal(g, g_ref); np.testing.assert_array_equal(a, a_ref)
        np.testing.assert_array_equal(aux, layouts.make_aux("redbluedoors", g_ref))
        np.testing.assert_array_equal(lw, ob.gen_words(lg))
    for seed in range(100):
        A, size = int(r.integers(1, 6)), int(r.integers(5, 12))
        lg = np.random.default_rng(5000 + seed); lw = ob.gen_words(lg)
        g_ref, a_ref = layouts.empty_layout(size, A, agent_start_pos=None, agent_start_dir=None, layout_rng=lg)
        np.testing.assert_array_equal(g_ref, layouts.empty_blank(size))
        g, a = ob.empty_random_layout(A, lw, layouts.empty_blank(size))
        np.testing.assert_array_equal(a, a_ref); np.testing.assert_array_equal(lw, ob.gen_words(lg))


CASES = [
    ("bup_a2", EnvSpec(11, 6, 2, 7, max_steps=7, joint_reward=True, env_kind="blockedunlockpickup"),
     dict(kind="blockedunlockpickup", room_size=6), 3000),
    ("bup_a4_rs8", EnvSpec(15, 8, 4, 7, max_steps=5, joint_reward=True, env_kind="blockedunlockpickup"),
     dict(kind="blockedunlockpickup", room_size=8), 1001),
    ("empty_random_9_a3", EnvSpec(9, 9, 3, 7, max_steps=6), dict(kind="empty_random"), 2000),
    ("empty_random_5_a6", EnvSpec(5, 5, 6, 5, max_steps=4), dict(kind="empty_random"), 515),
    ("empty_fixed_16_a4", EnvSpec(16, 16, 4, 7, max_steps=5), dict(kind="empty_fixed", start=(1, 1, 0)), 4099),
    ("rbd_8_a3", EnvSpec(16, 8, 3, 7, max_steps=6, joint_reward=True, failure_termination_mode="any", env_kind="redbluedoors"),
     dict(kind="redbluedoors"), 2500),
    ("rbd_6_a2", EnvSpec(12, 6, 2, 5, max_steps=4, joint_reward=True, failure_termination_mode="any", env_kind="redbluedoors"),
     dict(kind="redbluedoors"), 777),
    ("lh_4rooms_a2", EnvSpec(13, 9, 2, 7, max_steps=6, joint_reward=True, env_kind="lockedhallway"),
     dict(kind="lockedhallway", room_size=5, max_hallway_keys=1, max_keys_per_room=2), 1500),
    ("lh_8rooms_rs6_a3", EnvSpec(16, 21, 3, 7, max_steps=5, joint_reward=False, env_kind="lockedhallway"),
     dict(kind="lockedhallway", room_size=6, max_hallway_keys=2, max_keys_per_room=3), 600),
    ("playground_3x3_a3", EnvSpec(19, 19, 3, 7, max_steps=5), dict(kind="playground", room_size=7), 1000),
    ("playground_2x3_rs6_a2", EnvSpec(16, 11, 2, 5, max_steps=4), dict(kind="playground", room_size=6), 400),
    # rooms that are exactly FULL (round 6): the rejection sampling still ends, as in the reference
    ("bup_rs4_a2_full", EnvSpec(7, 4, 2, 3, max_steps=4, joint_reward=True, env_kind="blockedunlockpickup"),
     dict(kind="blockedunlockpickup", room_size=4), 300),
    ("bup_rs5_a7_full", EnvSpec(9, 5, 7, 3, max_steps=4, joint_reward=True, env_kind="blockedunlockpickup"),
     dict(kind="blockedunlockpickup", room_size=5), 200),
    ("empty_random_4_a3_full", EnvSpec(4, 4, 3, 3, max_steps=3), dict(kind="empty_random"), 300),
    ("rbd_4_a4_full", EnvSpec(8, 4, 4, 3, max_steps=3, joint_reward=True, failure_termination_mode="any", env_kind="redbluedoors"),
     dict(kind="redbluedoors"), 300),
]


def _run(env, spec, T, dev):
    g = torch.Generator(device=dev); g.manual_seed(5)
    outs = []
    for t in range(T):
        act = torch.randint(0, 7, (env.batch, spec.num_agents), dtype=torch.int8, device=dev, generator=g)
        o = env.step(act, auto_reset=True)
        outs.append([x.cpu().clone() for x in o] + [env.was_reset.cpu().clone()])
    return outs


def _make(spec, gen, B, dev, backend=None):
    env = BatchedMultiGridEnv(spec, B, dev, first_env=17, backend=backend)
    if spec.env_kind == "blockedunlockpickup":
        g0, a0, t0 = layouts.blockedunlockpickup_layout(gen["room_size"], spec.num_agents, np.random.default_rng(1), np.random.default_rng(2))
        env.load_state(g0, a0, aux=layouts.make_aux("blockedunlockpickup", g0, target=t0))
    elif spec.env_kind == "redbluedoors":
        g0, a0 = layouts.redbluedoors_layout(spec.height, spec.num_agents, np.random.default_rng(1))
        env.load_state(g0, a0, aux=layouts.make_aux("redbluedoors", g0))
    elif spec.env_kind == "lockedhallway":
        rs = gen["room_size"]
        g0, a0 = layouts.lockedhallway_layout(2 * ((spec.height - 1) // (rs - 1)), rs, 1, 2, spec.num_agents,
                                              np.random.default_rng(1), np.random.default_rng(2))
        env.load_state(g0, a0, aux=layouts.make_aux("lockedhallway", g0))
    elif gen["kind"] == "playground":
        rs = gen["room_size"]
        g0, a0 = layouts.playground_layout(rs, (spec.height - 1) // (rs - 1), (spec.width - 1) // (rs - 1), spec.num_agents,
                                           np.random.default_rng(1), np.random.default_rng(2))
        env.load_state(g0, a0)
    else:
        g0, a0 = layouts.empty_layout(spec.width, spec.num_agents)
        env.load_state(g0, a0)
    env.seed_synthetic(3)
    env.set_layout_generator(layout_seed=11, **gen)
    return env


@pytest.mark.gpu
@pytest.mark.parametrize("name,spec,gen,B", CASES, ids=[c[0] for c in CASES])
def test_device_generation_equals_oracle_over_many_episodes(name, spec, gen, B):
    dev = "cuda:0"
    T = 4 * spec.max_steps + 3
    hip = _make(spec, gen, B, dev)
    ref = _make(spec, gen, B, "cpu", backend=util.OracleBackend(spec, nthreads=8))
    got, want = _run(hip, spec, T, dev), None
    # the oracle side consumes the same actions (generated on the device, copied over)
    g = torch.Generator(device=dev); g.manual_seed(5)
    for t in range(T):
        act = torch.randint(0, 7, (B, spec.num_agents), dtype=torch.int8, device=dev, generator=g).cpu()
        o = [x.clone() for x in ref.step(act)]                  # step, then the finished envs are regenerated
        ref.reset_done()
        for k, (x, y) in enumerate(zip(got[t][:5], o)):
            assert torch.equal(x, y), f"{name} step {t} output {k}"
        assert torch.equal(got[t][5], ref.was_reset), f"{name} step {t} was_reset"
    for f in ("grid", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(hip, f).cpu(), getattr(ref, f)), f
    assert torch.equal(hip._gen["gen_state"].cpu(), ref._gen["gen_state"])
    assert int(hip.episode.sum()) >= 3 * B
    hip.check_errors()
    # the fused launch (mgx_step_generate) == the two launches (mgx_step, mgx_reset_generate), also with the one-hot output
    two = _make(spec, gen, B, dev)
    g = torch.Generator(device=dev); g.manual_seed(5)
    for t in range(T):
        act = torch.randint(0, 7, (B, spec.num_agents), dtype=torch.int8, device=dev, generator=g)
        o = two.step(act, one_hot=(t % 2 == 1))
        two.reset_done()
        if t % 2 == 0:
            assert torch.equal(o[0].cpu(), got[t][0]), f"{name} two-launch step {t}"
        assert torch.equal(two.was_reset.cpu(), got[t][5])
    for f in ("grid", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(hip, f), getattr(two, f)), f


@pytest.mark.gpu
@pytest.mark.parametrize("one_hot", [False, True])
def test_rollout_with_device_generation_equals_steps(one_hot):
    """mgx_step_ex(steps = T, generate): the rollout form of the step with episode starts generated on the device == T calls of
    step(auto_reset=True) -- every output slice, `was_reset[T,B]`, the whole state and the generator states (the mode matrix:
    rollout x generation, also with the one-hot output)."""
    name, spec, gen, B = CASES[0]
    dev = "cuda:0"
    T = 2 * spec.max_steps + 5
    a, b = _make(spec, gen, B, dev), _make(spec, gen, B, dev)
    g = torch.Generator(device=dev); g.manual_seed(5)
    acts = torch.randint(0, 7, (T, B, spec.num_agents), dtype=torch.int8, device=dev, generator=g)
    out = b.rollout(acts, auto_reset=True, one_hot=one_hot)
    for t in range(T):
        o = a.step(acts[t], auto_reset=True, one_hot=one_hot)
        for k, key in enumerate(("obs", "dir", "reward", "terminated", "truncated")):
            assert torch.equal(out[key][t], o[k]), f"{name} step {t} {key}"
        assert torch.equal(out["was_reset"][t], a.was_reset), f"{name} step {t} was_reset"
    for f in ("grid", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    assert torch.equal(a._gen["gen_state"], b._gen["gen_state"])
    assert int(b.episode.sum()) >= B
    b.check_errors()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["candidates", "between", "side"])
@pytest.mark.parametrize("case", [0, 2, 5], ids=lambda k: CASES[k][0])
def test_generator_launches_of_their_own_equal_unstaged(case, mode):
    """MgxGenStage.external (set_layout_generator(staged="candidates" -- the default -- / "between" / "side")): the staging slots are
    filled by launches of their own (mgx_stage_generate) between the steps or on a stream beside them -- one candidate per value of
    the generator's np_random draw while the current episode runs ("candidates": every episode end adopts), or `lead` steps ahead
    of each truncation.  Bit-identical to generating every episode in the tail of its step -- eagerly, as a captured hipGraph and
    as a rollout -- and the slots ARE used (adoptions happen)."""
    name, spec, gen, B = CASES[case]
    if spec.max_steps < 6:
        spec = EnvSpec(**{**spec.as_dict(), "max_steps": 9})
    dev = "cuda:0"
    T = 5 * spec.max_steps + 2
    g = torch.Generator(device=dev); g.manual_seed(3)
    acts = torch.randint(0, 7, (T, B, spec.num_agents), dtype=torch.int8, device=dev, generator=g)

    def make(staged, lead=None):
        env = _make(spec, gen, B, dev)
        env.set_layout_generator(layout_seed=11, staged=staged, lead=lead, **gen)
        env.step_count.copy_(torch.arange(B, device=dev, dtype=torch.int32) % spec.max_steps)      # episodes out of phase
        return env

    ref, side = make(False), make(mode, lead=4)
    assert side._gen["stage"]["external"] and side._gen["stage"]["lead"] == 4
    assert bool(side._gen["stage"].get("candidates")) == (mode == "candidates")
    assert (side._gen["stage"]["stream"] is not None) == (mode == "side")
    for t in range(T):
        want = [x.clone() for x in ref.step(acts[t], auto_reset=True)] + [ref.was_reset.clone()]
        got = list(side.step(acts[t], auto_reset=True)) + [side.was_reset]
        for k, (w, gg) in enumerate(zip(want, got)):
            assert torch.equal(w, gg), f"{name} step {t} output {k}"
    torch.cuda.synchronize()
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(ref, f), getattr(side, f)), f
    assert torch.equal(ref._gen["gen_state"], side._gen["gen_state"])
    served = int((side._gen["stage"]["tag"][:, 0] >= 0).sum())
    assert served > B // 2, served                                   # the generator launches did fill the slots
    # ... and as a graph: the generator launches are parallel branches of the captured block
    ref2, side2 = make(False), make(mode, lead=4)
    K = 2 * spec.max_steps
    graph = side2.capture_steps(acts[:K], auto_reset=True)
    for rep in range(2):
        graph.replay()
        for t in range(K):
            ref2.step(acts[t], auto_reset=True)
    torch.cuda.synchronize()
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode", "obs", "reward"):
        assert torch.equal(getattr(ref2, f), getattr(side2, f)), f"graph: {f}"
    side.check_errors(); side2.check_errors()
    # the rollout form issues the generator launches inside mgx_step_ex (T launches over the [t] slices: they must be 16-byte
    # aligned, include/mgx.h)
    if mode in ("between", "candidates") and (B * spec.num_agents * spec.view_size ** 2 * 3) % 16 == 0:
        ref3, roll = make(False), make(mode, lead=4)
        out = roll.rollout(acts[:K], auto_reset=True)
        for t in range(K):
            o = ref3.step(acts[t], auto_reset=True)
            assert torch.equal(out["obs"][t], o[0]) and torch.equal(out["was_reset"][t], ref3.was_reset), f"rollout step {t}"
        for f in ("cells", "agents", "rng", "step_count", "aux", "episode"):
            assert torch.equal(getattr(ref3, f), getattr(roll, f)), f"rollout: {f}"
        assert int((roll._gen["stage"]["tag"][:, 0] >= 0).sum()) > B // 2


@pytest.mark.gpu
def test_every_episode_end_adopts_its_candidate():
    """MgxGenStage.candidates: an episode that ends EARLY -- here by success: an agent is handed the target box, the
    BlockedUnlockPickup hook fires on the next step -- adopts the candidate its door-row draw selects, exactly like a truncation.
    (a) staged == unstaged bit for bit through early ends, truncations and the steps after them; (b) the slots really are what
    the envs restart from: with every candidate grid overwritten by a marker after the generator launch, the envs that restart
    -- early-ended and truncated alike -- come out holding the marker."""
    name, spec, gen, B = CASES[0]
    spec = EnvSpec(**{**spec.as_dict(), "max_steps": 24})
    B = 1024
    dev = "cuda:0"
    T = 3 * spec.max_steps
    g = torch.Generator(device=dev); g.manual_seed(4)
    acts = torch.randint(0, 7, (T, B, spec.num_agents), dtype=torch.int8, device=dev, generator=g)

    def make(staged):
        env = _make(spec, gen, B, dev)
        env.set_layout_generator(layout_seed=11, staged=staged, lead=4, **gen)
        env.step_count.copy_(torch.arange(B, device=dev, dtype=torch.int32) % 5)                # a little out of phase
        return env

    def hand_over_the_box(env, which):
        """agent 0 of the envs `which` carries the target box from now on (written in place: the state is not 'replaced')"""
        a = env.agents                                                                          # u8[B, A, 8]: bytes 5..6 = carrying
        a[which, 0, 5] = 7                                                                      # Type.box
        a[which, 0, 6] = env.aux[which, 1]                                                      # the target's colour (include/mgx.h)

    early = torch.arange(B, device=dev) % 3 == 0
    ref, cand = make(False), make("candidates")
    assert cand._gen["stage"]["candidates"] == 4
    n_early = 0
    for t in range(T):
        if t % 9 == 6:                                       # (at least two generator launches after the last restart)
            hand_over_the_box(ref, early); hand_over_the_box(cand, early)
        want = [x.clone() for x in ref.step(acts[t], auto_reset=True)] + [ref.was_reset.clone()]
        got = list(cand.step(acts[t], auto_reset=True)) + [cand.was_reset]
        for k, (w, gg) in enumerate(zip(want, got)):
            assert torch.equal(w, gg), f"step {t} output {k}"
        n_early += int((cand.was_reset.bool() & ~cand.truncated.bool()).sum())
    torch.cuda.synchronize()
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(ref, f), getattr(cand, f)), f
    assert torch.equal(ref._gen["gen_state"], cand._gen["gen_state"])
    assert n_early > B // 3, n_early                          # success-ended episodes did occur
    cand.check_errors()
    # (b) poisoned candidates show up in the restarted envs
    env = make("candidates")
    marker = 0x0103                                           # (a valid cell: a floor of colour 1)
    for t in range(8):
        env.step(acts[t], auto_reset=True)
    st = env._gen["stage"]
    ready = (st["tag"] == env.episode[:, None]).all(dim=1)    # every candidate of the env's current episode is there
    assert int(ready.sum()) > B // 2
    st["grid"].fill_(marker)
    hand_over_the_box(env, early)
    env.step(acts[8], auto_reset=True)
    restarted = env.was_reset.bool()
    took = (env.cells.reshape(B, -1) == marker).all(dim=1)
    assert int((restarted & early).sum()) > B // 4 and int((restarted & ~early).sum()) >= 0
    assert torch.equal(took & ready, restarted & ready), "an env whose candidates were ready restarted from somewhere else"
    assert not bool((took & ~restarted).any())


@pytest.mark.gpu
def test_candidates_on_sub_shard_chains():
    """step(sub_shards=P) (mgx_step_chains: P launches on P streams) and capture_steps(sub_shards=P) with the candidates protocol:
    the chains are joined every lead/2 steps for ONE generator launch over the whole batch (eager form), the split shards of a
    captured graph make their own; both equal the unstaged run bit for bit, and the candidates are made and adopted."""
    name, spec, gen, B = CASES[0]
    spec = EnvSpec(**{**spec.as_dict(), "max_steps": 11})
    B, dev = 1000, "cuda:0"
    T = 4 * spec.max_steps + 1
    g = torch.Generator(device=dev); g.manual_seed(12)
    acts = torch.randint(0, 7, (T, B, spec.num_agents), dtype=torch.int8, device=dev, generator=g)

    def make(staged):
        env = _make(spec, gen, B, dev)
        env.set_layout_generator(layout_seed=11, staged=staged, lead=4, **gen)
        env.step_count.copy_(torch.arange(B, device=dev, dtype=torch.int32) % spec.max_steps)
        return env

    ref, cand = make(False), make("candidates")
    for t in range(T):
        want = [x.clone() for x in ref.step(acts[t], auto_reset=True)] + [ref.was_reset.clone()]
        got = list(cand.step(acts[t], auto_reset=True, sub_shards=3)) + [cand.was_reset]
        cand.join()
        for k, (w, gg) in enumerate(zip(want, got)):
            assert torch.equal(w, gg), f"step {t} output {k}"
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(ref, f), getattr(cand, f)), f
    assert torch.equal(ref._gen["gen_state"], cand._gen["gen_state"])
    st = cand._gen["stage"]
    assert st.get("launches", 0) >= T // 2 - 1 and int((st["tag"] == cand.episode[:, None]).all(dim=1).sum()) > B // 2
    # the captured form: every split shard makes the candidates of its own slice
    ref2, cand2 = make(False), make("candidates")
    K = 2 * spec.max_steps
    graph = cand2.capture_steps(acts[:K], auto_reset=True, sub_shards=2)
    for rep in range(2):
        graph.replay()
        for t in range(K):
            ref2.step(acts[t], auto_reset=True)
    torch.cuda.synchronize()
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode"):
        assert torch.equal(getattr(ref2, f), getattr(cand2, f)), f"graph: {f}"
    assert int((cand2._gen["stage"]["tag"] == cand2.episode[:, None]).all(dim=1).sum()) > B // 2
    cand.check_errors(); cand2.check_errors()


@pytest.mark.gpu
def test_a_captured_block_shorter_than_the_generator_cadence_still_makes_candidates():
    """capture_steps() of fewer steps than lie between two generator launches (set_layout_generator's default cadence is a third of
    max_steps): the block would contain no generator launch at all and its replays would never make a candidate -- every episode
    end generated in the tail, silently slow.  The captured block ends with one; results as unstaged either way."""
    name, spec, gen, B = CASES[0]
    spec = EnvSpec(**{**spec.as_dict(), "max_steps": 40})
    B, dev, K = 512, "cuda:0", 8
    g = torch.Generator(device=dev); g.manual_seed(6)
    acts = torch.randint(0, 7, (K, B, spec.num_agents), dtype=torch.int8, device=dev, generator=g)

    def make(staged):
        env = _make(spec, gen, B, dev)
        env.set_layout_generator(layout_seed=11, staged=staged, **gen)
        env.step_count.copy_(torch.arange(B, device=dev, dtype=torch.int32) % spec.max_steps)
        return env

    ref, cand = make(False), make("candidates")
    st = cand._gen["stage"]
    assert st["candidates"] == 4 and st["lead"] // 2 > K              # no launch of the regular cadence falls into K steps
    graph = cand.capture_steps(acts, auto_reset=True)
    for rep in range(12):
        graph.replay()
        for t in range(K):
            ref.step(acts[t], auto_reset=True)
    torch.cuda.synchronize()
    for f in ("cells", "agents", "rng", "step_count", "aux", "episode", "obs", "reward"):
        assert torch.equal(getattr(ref, f), getattr(cand, f)), f
    assert torch.equal(ref._gen["gen_state"], cand._gen["gen_state"])
    ready = (st["tag"] == cand.episode[:, None]).all(dim=1)
    assert int(ready.sum()) > B * 3 // 4, int(ready.sum())              # the candidates ARE being made
    assert int(cand.episode.sum()) > B                                   # ... and episodes did end
    cand.check_errors()


@pytest.mark.gpu
@pytest.mark.parametrize("protocol", ["candidates", "between"])
def test_staged_slots_are_a_cache_not_state(protocol):
    """The staging slots of the truncation resets (include/mgx.h: MgxGenStage) must never change results: staged == unstaged bit for
    bit when (a) the caller replaces np_random / the step counts mid-episode AFTER the snapshot was taken (seed_synthetic, a
    partial load of step counts), and (b) a truncated env is stepped once more WITHOUT auto-reset and only then with it (the slot
    was made for np_random as it was at the truncation step)."""
    name, spec, gen, B = CASES[0]
    dev = "cuda:0"
    M = spec.max_steps
    g = torch.Generator(device=dev); g.manual_seed(9)
    acts = torch.randint(0, 7, (3 * M, B, spec.num_agents), dtype=torch.int8, device=dev, generator=g)

    def run(staged, scenario):
        env = _make(spec, gen, B, dev)
        env.set_layout_generator(layout_seed=11, staged=staged, **gen)
        outs = []
        for t in range(2 * M + 4):
            if scenario == "reseed" and t == M - 1:              # the snapshot (step M - 2) is taken, its slot generated: now
                env.seed_synthetic(77)                           # np_random is replaced
            if scenario == "step_past" and t == M - 1:
                o = env.step(acts[t], auto_reset=False)          # the truncating step, without auto-reset ...
                outs.append([x.clone() for x in o])
                continue                                         # ... the envs are regenerated one step LATER
            o = env.step(acts[t], auto_reset=True)
            outs.append([x.clone() for x in o] + [env.was_reset.clone()])
        env.check_errors()
        return outs, {f: getattr(env, f).clone() for f in ("cells", "agents", "rng", "step_count", "aux", "episode")}, \
            env._gen["gen_state"].clone()

    for scenario in ("reseed", "step_past"):
        (o1, s1, g1), (o2, s2, g2) = run(protocol, scenario), run(False, scenario)
        for t, (a, b) in enumerate(zip(o1, o2)):
            for k, (x, y) in enumerate(zip(a, b)):
                assert torch.equal(x, y), f"{scenario}: step {t} output {k}"
        for f in s1:
            assert torch.equal(s1[f], s2[f]), f"{scenario}: {f}"
        assert torch.equal(g1, g2), scenario
        assert int(s1["episode"].sum()) >= B


@pytest.mark.gpu
def test_generated_bup_episodes_have_the_reference_structure():
    dev = "cuda:0"
    spec = EnvSpec(11, 6, 2, 7, max_steps=1, joint_reward=True, env_kind="blockedunlockpickup")
    B = 20000
    env = _make(spec, dict(kind="blockedunlockpickup", room_size=6), B, dev)
    env.step(torch.zeros((B, 2), dtype=torch.int8, device=dev))          # truncates every env
    assert int(env.reset_done().sum()) == B
    g, a, aux = env.grid.cpu().numpy(), env.agents.cpu().numpy(), env.aux.cpu().numpy()
    t = g[..., 0]
    assert ((t == 7).sum(axis=(1, 2)) == 1).all() and (t[:, :, 6:] == 7).sum() == B          # one box, right room
    door = (t == 4)
    assert (door.sum(axis=(1, 2)) == 1).all() and door[:, 1:5, 5].sum() == B and (g[..., 2][door] == 2).all()   # locked, in the wall
    ys = door[:, :, 5].argmax(axis=1)
    assert (t[np.arange(B), ys, 4] == 6).all()                                               # ball in front of it
    key = (t == 5)
    assert (key.sum(axis=(1, 2)) == 1).all() and key[:, :, :5].sum() == B
    assert (g[..., 1][key] == g[..., 1][door]).all()                                          # the key fits the door
    assert (a[:, :, 2] >= 1).all() and (a[:, :, 2] <= 4).all() and (a[:, :, 4] == 0).all()    # agents in the left room, alive
    np.testing.assert_array_equal(aux[:, 0], 7); np.testing.assert_array_equal(aux[:, 1], g[..., 1][t == 7])
    # all four door rows and all six colours occur: the draws are not degenerate
    assert len(np.unique(ys)) == 4 and len(np.unique(aux[:, 1])) == 6
    assert len({r.tobytes() for r in g[:2000]}) > 1500


def test_generator_state_round_trip_on_cpu():
    spec = EnvSpec(11, 6, 2, 7, max_steps=3, joint_reward=True, env_kind="blockedunlockpickup")
    gen = dict(kind="blockedunlockpickup", room_size=6)
    a = _make(spec, gen, 40, "cpu", backend=util.OracleBackend(spec))
    acts = [torch.from_numpy(util.random_actions(40, 2, t, p_missing=0)) for t in range(14)]
    for t in range(6):
        a.step(acts[t], auto_reset=True)
    sd = a.state_dict()
    want = [[x.clone() for x in a.step(acts[t], auto_reset=True)] for t in range(6, 14)]
    b = BatchedMultiGridEnv(spec, 40, "cpu", first_env=17, backend=util.OracleBackend(spec))
    b.load_state_dict(sd)
    for t in range(6, 14):
        for x, y in zip(want[t - 6], b.step(acts[t], auto_reset=True)):
            assert torch.equal(x, y)
    assert torch.equal(a.grid, b.grid) and torch.equal(a._gen["gen_state"], b._gen["gen_state"])


def test_staging_protocol_chosen_by_the_generator_kind_on_cpu():
    """set_layout_generator(staged=True): `candidates` (one slot per value of the generator's single np_random draw, include/mgx.h:
    MgxGenStage.candidates) for the kinds that draw from env.np_random at most once and offer at most four values; the snapshot
    protocol otherwise; the slot tensors' shapes; the tags all -1 (nothing staged) -- and dropped again when the state is replaced."""
    def stage_of(spec, gen, **kw):
        env = BatchedMultiGridEnv(spec, 8, "cpu", backend=util.OracleBackend(spec))
        env.set_layout_generator(layout_seed=1, **gen, **kw)
        return env, env._gen.get("stage")
    bup = EnvSpec(11, 6, 2, 7, max_steps=90, joint_reward=True, env_kind="blockedunlockpickup")
    env, st = stage_of(bup, dict(kind="blockedunlockpickup", room_size=6))
    assert st["candidates"] == 4 and st["external"] == 2 and st["lead"] == 60                 # a launch every max_steps / 3 steps
    assert tuple(st["grid"].shape) == (8, 4, 6, 11) and tuple(st["agents"].shape) == (8, 4, 2, 8)
    assert tuple(st["aux"].shape) == (8, 4, 16) and tuple(st["words"].shape) == (8, 4, 6) and bool((st["tag"] == -1).all())
    st["tag"].fill_(5); env.seed_synthetic(3)
    assert bool((st["tag"] == -1).all())                                                      # the slots are a cache: dropped
    _, st = stage_of(EnvSpec(15, 8, 2, 7, max_steps=90, joint_reward=True, env_kind="blockedunlockpickup"),
                     dict(kind="blockedunlockpickup", room_size=8))
    assert not st.get("candidates") and st["external"] == 2                                   # 6 door rows: the snapshot protocol
    for spec, gen in ((EnvSpec(9, 9, 3, 7, max_steps=30), dict(kind="empty_random")),
                      (EnvSpec(16, 8, 3, 7, max_steps=30, joint_reward=True, failure_termination_mode="any", env_kind="redbluedoors"),
                       dict(kind="redbluedoors")),
                      (EnvSpec(13, 9, 2, 7, max_steps=30, joint_reward=True, env_kind="lockedhallway"),
                       dict(kind="lockedhallway", room_size=5))):
        _, st = stage_of(spec, gen)
        assert st["candidates"] == 1 and tuple(st["grid"].shape) == (8, 1, spec.height, spec.width)
    _, st = stage_of(EnvSpec(19, 19, 3, 7, max_steps=30), dict(kind="playground", room_size=7))
    assert not st.get("candidates")                                                           # many np_random draws
    with pytest.raises(ValueError, match="candidates"):
        stage_of(EnvSpec(19, 19, 3, 7, max_steps=30), dict(kind="playground", room_size=7), staged="candidates")
    _, st = stage_of(EnvSpec(9, 9, 1, 7, max_steps=30), dict(kind="empty_random"))
    assert st is None                                                                         # one agent: no staging at all


def test_generators_refuse_layouts_without_room_to_spare():
    """The reference's place_obj samples until a position fits, without a bound (base.py:604-669): in a room too small for what goes
    into it, it never returns -- on a GPU that lane would hang the device.  The C ABI refuses such specs before any launch
    (MGX_ERR_UNSUPPORTED from mgx_reset_generate; no GPU needed: the check precedes the launch)."""
    import ctypes as C
    from multigrid_amd import _lib
    L = _lib.lib()

    def rc_of(spec, kind, room_size=0, max_hallway_keys=1, max_keys_per_room=2, aux=4096):
        sc = spec.to_c()
        g = _lib.MgxLayoutGen(_lib.GEN_KINDS[kind], room_size, 1, 1, 0, max_hallway_keys, max_keys_per_room, 4096, 4096)
        fake = 4096                                              # (aligned, never dereferenced: the checks come first)
        return L.mgx_reset_generate(C.byref(sc), 8, C.byref(g), fake, fake, fake, fake, aux, fake, None, None)

    bup = lambda rs, A: EnvSpec(2 * rs - 1, rs, A, 7, max_steps=9, joint_reward=True, env_kind="blockedunlockpickup")
    rbd = lambda A: EnvSpec(8, 4, A, 3, max_steps=9, joint_reward=True, failure_termination_mode="any", env_kind="redbluedoors")
    lh = lambda rows, A: EnvSpec(10, 3 * rows + 1, A, 3, max_steps=9, joint_reward=True, env_kind="lockedhallway")
    assert rc_of(bup(4, 3), "blockedunlockpickup", 4) == _lib.ERR_UNSUPPORTED        # 2x2 room: key + ball + 3 agents do not fit
    assert rc_of(bup(5, 8), "blockedunlockpickup", 5) == _lib.ERR_UNSUPPORTED
    assert rc_of(EnvSpec(9, 9, 4, 7, max_steps=9), "playground", 5) == _lib.ERR_UNSUPPORTED      # 12 objects may draw one 3x3 room
    assert rc_of(EnvSpec(11, 11, 4, 7, max_steps=9), "playground", 6) == _lib.ERR_UNSUPPORTED
    assert rc_of(EnvSpec(4, 4, 4, 3, max_steps=9), "empty_random") == _lib.ERR_UNSUPPORTED        # 2x2 interior: goal + 4 agents
    assert rc_of(rbd(5), "redbluedoors") == _lib.ERR_UNSUPPORTED                                  # a 2x2 middle room for 5 agents
    assert rc_of(lh(1, 4), "lockedhallway", 4, 1, 2) == _lib.ERR_UNSUPPORTED                      # a 2x2 hallway: a key + 4 agents
    assert rc_of(lh(2, 2), "lockedhallway", 4, 1, 5) == _lib.ERR_UNSUPPORTED                      # 5 keys into a 2x2 room
    assert rc_of(lh(4, 2), "lockedhallway", 4, 1, 3) == _lib.ERR_UNSUPPORTED                      # 8 rooms: a colour comes up twice
    # ... and a room that is exactly full is NOT refused: the feasibility check passes, and the call then stops at the argument test
    # behind it -- a hook env without its `aux` -- before any launch (Empty-Random's exact fit: the GPU case `empty_random_4_a3_full`)
    for rc in (rc_of(bup(4, 2), "blockedunlockpickup", 4, aux=None), rc_of(bup(5, 7), "blockedunlockpickup", 5, aux=None),
               rc_of(rbd(4), "redbluedoors", aux=None), rc_of(lh(1, 3), "lockedhallway", 4, 1, 2, aux=None),
               rc_of(lh(2, 2), "lockedhallway", 4, 1, 4, aux=None)):
        assert rc == _lib.ERR_INVALID_ARGUMENT


def test_oracle_shuffle_equals_numpy():
    """numpy's Generator.shuffle of a Python list (RandomMixin._rand_perm, multigrid/utils/random.py:75-83) restated."""
    r = np.random.default_rng(5)
    for trial in range(300):
        g = np.random.Generator(np.random.PCG64(int(r.integers(0, 2 ** 40))))
        if r.random() < 0.5:
            g.integers(0, 5)                                   # leaves a buffered 32-bit half behind
        w = ob.gen_words(g)
        n = int(r.integers(1, 25))
        items = [int(v) for v in r.integers(0, 6, size=n)]
        want = list(items); g.shuffle(want)
        assert ob.np_shuffle(w, items) == want
        np.testing.assert_array_equal(w, ob.gen_words(g))


def test_layout_oracle_equals_layouts_py_lockedhallway_and_playground():
    r = np.random.default_rng(9)
    for seed in range(250):
        # (feasible parameters only: the hallway must hold its keys and the agents, a room its keys -- the reference's rejection
        # sampling loops forever otherwise, base.py:640-669 with max_tries = inf)
        n = int(r.choice([2, 4, 6, 8, 12, 16])); rs = int(r.integers(5, 8)); A = int(r.integers(1, 4))
        mhk, mkr = int(r.integers(1, 3)), int(r.integers(1, 4))
        lg = np.random.default_rng(8000 + seed)
        if seed % 3 == 0:
            lg.integers(0, 7)
        lw = ob.gen_words(lg)
        g_ref, a_ref = layouts.lockedhallway_layout(n, rs, mhk, mkr, A, lg, np.random.default_rng(1))
        g, a, aux = ob.lh_layout(n, rs, mhk, mkr, A, lw, layouts.lockedhallway_blank(n, rs))
        np.testing.assert_array_equal(g, g_ref, err_msg=f"seed {seed} n {n} rs {rs}"); np.testing.assert_array_equal(a, a_ref)
        np.testing.assert_array_equal(aux, layouts.make_aux("lockedhallway", g_ref))
        np.testing.assert_array_equal(lw, ob.gen_words(lg))
    for seed in range(150):
        rs, rows, cols, A = int(r.integers(6, 9)), int(r.integers(2, 4)), int(r.integers(2, 4)), int(r.integers(1, 5))
        lg, ng = np.random.default_rng(9000 + seed), np.random.default_rng(9500 + seed)
        if seed % 3 == 0:
            ng.integers(0, 7)
        lw, nw = ob.gen_words(lg), ob.gen_words(ng)
        g_ref, a_ref = layouts.playground_layout(rs, rows, cols, A, lg, ng)
        g, a = ob.playground_layout(rs, rows, cols, A, lw, nw, layouts.roomgrid_blank(rs, rows, cols))
        np.testing.assert_array_equal(g, g_ref, err_msg=f"seed {seed}"); np.testing.assert_array_equal(a, a_ref)
        np.testing.assert_array_equal(lw, ob.gen_words(lg)); np.testing.assert_array_equal(nw, ob.gen_words(ng))
