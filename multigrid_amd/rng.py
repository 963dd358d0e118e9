"""PCG64 state plumbing between numpy (host) and the device `rng` tensor (u64[B,4], include/mgx.h).

`gymnasium.Env.reset(seed=s)` sets `np_random = Generator(PCG64(SeedSequence(s)))`; the reference's
`handle_actions` draws `np_random.random(size=A)` from it every step (multigrid/base.py:396-399).  The device
kernel continues exactly that stream, so all the host has to do is hand over the 128-bit state and increment.
"""
from __future__ import annotations

import numpy as np

M64 = (1 << 64) - 1


def words_from_bitgen_state(state: dict) -> np.ndarray:
    """numpy `PCG64().state` dict -> u64[4] = [state_lo, state_hi, inc_lo, inc_hi]."""
    st = state["state"]
    s, inc = int(st["state"]), int(st["inc"])
    return np.array([s & M64, s >> 64, inc & M64, inc >> 64], dtype=np.uint64)


def bitgen_state_from_words(words) -> dict:
    w = [int(x) for x in np.asarray(words, dtype=np.uint64)]
    return {"bit_generator": "PCG64",
            "state": {"state": w[0] | (w[1] << 64), "inc": w[2] | (w[3] << 64)},
            "has_uint32": 0, "uinteger": 0}


def generator_from_words(words) -> np.random.Generator:
    bg = np.random.PCG64()
    bg.state = bitgen_state_from_words(words)
    return np.random.Generator(bg)


def seeded_generator(seed) -> np.random.Generator:
    """What `gymnasium.utils.seeding.np_random(seed)` returns."""
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


def words_from_seed(seed) -> np.ndarray:
    return words_from_bitgen_state(np.random.PCG64(np.random.SeedSequence(seed)).state)


def words_from_seeds(seeds) -> np.ndarray:
    """Per-env states for per-env integer seeds: env b gets Generator(PCG64(SeedSequence(seeds[b])))."""
    seeds = np.asarray(seeds)
    out = np.empty((len(seeds), 4), dtype=np.uint64)
    for b, s in enumerate(seeds):
        out[b] = words_from_seed(int(s))
    return out


def words_from_seed_and_index(seed: int, global_index) -> np.ndarray:
    """Env with global index g gets Generator(PCG64(SeedSequence([seed, g]))); g == 0 gets SeedSequence(seed), the
    reference's single env (multigrid/base.py:269)."""
    idx = np.asarray(global_index)
    out = np.empty((len(idx), 4), dtype=np.uint64)
    for b, g in enumerate(idx):
        out[b] = words_from_seed(int(seed) if int(g) == 0 else [int(seed), int(g)])
    return out


def synthetic_words(batch: int, seed: int, first_env: int = 0) -> np.ndarray:
    """Cheap valid PCG64 states for benchmarks (any 128-bit state, odd increment), a pure function of the
    GLOBAL env index so that sharding the batch over ranks does not change any env's stream."""
    idx = np.arange(first_env, first_env + batch, dtype=np.uint64)
    out = np.empty((batch, 4), dtype=np.uint64)
    x = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)
    for k in range(4):      # splitmix64 per word
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        out[:, k] = z ^ (z >> np.uint64(31))
    out[:, 2] |= np.uint64(1)
    return out


def gen_words_from_generator(gen: np.random.Generator) -> np.ndarray:
    """numpy Generator(PCG64) -> u64[5] = [state_lo, state_hi, inc_lo, inc_hi, has_uint32 << 32 | uinteger]: the PCG64 words
    plus numpy's 32-bit output buffer, which `Generator.integers` draws through (include/mgx.h: gen_state)."""
    st = gen.bit_generator.state
    w = words_from_bitgen_state(st)
    return np.concatenate([w, np.array([(int(st["has_uint32"]) << 32) | int(st["uinteger"])], dtype=np.uint64)])


def generator_from_gen_words(words5) -> np.random.Generator:
    w = [int(x) for x in np.asarray(words5, dtype=np.uint64)]
    bg = np.random.PCG64()
    st = bitgen_state_from_words(w[:4])
    st["has_uint32"], st["uinteger"] = w[4] >> 32, w[4] & 0xFFFFFFFF
    bg.state = st
    return np.random.Generator(bg)


def layout_gen_state(layout_seed: int, global_index) -> np.ndarray:
    """gen_state u64[B,6] for mgx_reset_generate: env g's placement generator is Generator(PCG64(SeedSequence([layout_seed,
    g]))) (the reference seeds it from OS entropy at construction), empty 32-bit buffers."""
    idx = np.asarray(global_index)
    out = np.zeros((len(idx), 6), dtype=np.uint64)
    for b, g in enumerate(idx):
        out[b, :4] = words_from_seed([int(layout_seed), int(g)])
    return out
