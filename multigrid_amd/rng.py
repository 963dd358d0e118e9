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


# ---- numpy's SeedSequence -> PCG64 seeding, vectorised over envs (numpy/random/bit_generator.pyx: SeedSequence.mix_entropy /
# generate_state, pool_size 4; numpy/random/src/pcg64/pcg64.h: pcg_setseq_128_srandom_r).  Pinned against numpy itself in
# tests/test_env_compat.py; a batch of 1M envs is seeded in a fraction of a second instead of one SeedSequence object per env.
_XSHIFT = np.uint32(16)
_INIT_A, _MULT_A = 0x43B0D7E5, 0x931E8875
_INIT_B, _MULT_B = 0x8B51F9DD, 0x58F38DED
_MIX_L, _MIX_R = np.uint32(0xCA01F9DD), np.uint32(0x4973F715)
_PCG_MULT = (0x2360ED051FC65DA4 << 64) | 0x4385DF649FCCF645


def _seed_sequence_pool(entropy_words) -> list:
    """SeedSequence(entropy).pool for a batch: `entropy_words` = list of u32[B] arrays (the assembled entropy, word by word)."""
    const = [_INIT_A]

    def hashmix(v):
        v = v ^ np.uint32(const[0])
        const[0] = (const[0] * _MULT_A) & 0xFFFFFFFF
        v = v * np.uint32(const[0])
        return v ^ (v >> _XSHIFT)

    def mix(x, y):
        r = _MIX_L * x - _MIX_R * y
        return r ^ (r >> _XSHIFT)

    n = len(entropy_words)
    zero = np.zeros_like(entropy_words[0])
    pool = [hashmix(entropy_words[i] if i < n else zero) for i in range(4)]
    for i_src in range(4):
        for i_dst in range(4):
            if i_src != i_dst:
                pool[i_dst] = mix(pool[i_dst], hashmix(pool[i_src]))
    for i_src in range(4, n):
        for i_dst in range(4):
            pool[i_dst] = mix(pool[i_dst], hashmix(entropy_words[i_src]))
    return pool


def _generate_state_u64x4(pool) -> list:
    """SeedSequence.generate_state(4, np.uint64) for a batch -> four u64[B] arrays."""
    const, out = _INIT_B, []
    for i in range(8):
        v = pool[i % 4] ^ np.uint32(const)
        const = (const * _MULT_B) & 0xFFFFFFFF
        v = v * np.uint32(const)
        out.append((v ^ (v >> _XSHIFT)).astype(np.uint64))
    return [out[2 * k] | (out[2 * k + 1] << np.uint64(32)) for k in range(4)]


def _mul128(a_lo, a_hi, b_lo: int, b_hi: int):
    """(a * b) mod 2^128 for u64[B] halves of a and a Python-int constant b, in 32-bit limbs (no overflow in uint64)."""
    m32 = np.uint64(0xFFFFFFFF)
    a = [a_lo & m32, a_lo >> np.uint64(32), a_hi & m32, a_hi >> np.uint64(32)]
    b = [np.uint64((b_lo >> 0) & 0xFFFFFFFF), np.uint64(b_lo >> 32), np.uint64(b_hi & 0xFFFFFFFF), np.uint64(b_hi >> 32)]
    limbs, carry = [], np.zeros_like(a_lo)
    for k in range(4):
        acc_lo, acc_hi = carry & m32, carry >> np.uint64(32)            # column sum kept as two 32-bit-safe accumulators
        for i in range(k + 1):
            p = a[i] * b[k - i]                                           # < 2^64
            acc_lo = acc_lo + (p & m32)
            acc_hi = acc_hi + (p >> np.uint64(32))
        acc_hi = acc_hi + (acc_lo >> np.uint64(32))
        limbs.append(acc_lo & m32)
        carry = acc_hi
    return limbs[0] | (limbs[1] << np.uint64(32)), limbs[2] | (limbs[3] << np.uint64(32))


def _add128(a_lo, a_hi, b_lo, b_hi):
    lo = a_lo + b_lo
    return lo, a_hi + b_hi + (lo < a_lo).astype(np.uint64)


def _pcg64_words_from_entropy(entropy_words) -> np.ndarray:
    """PCG64(SeedSequence(entropy)) for a batch -> u64[B,4] = [state_lo, state_hi, inc_lo, inc_hi]."""
    with np.errstate(over="ignore"):
        w = _generate_state_u64x4(_seed_sequence_pool(entropy_words))
        init_hi, init_lo, seq_hi, seq_lo = w                             # pcg64_set_seed: seed = {high, low}, inc = {high, low}
        inc_lo = (seq_lo << np.uint64(1)) | np.uint64(1)
        inc_hi = (seq_hi << np.uint64(1)) | (seq_lo >> np.uint64(63))
        m_lo, m_hi = _PCG_MULT & M64, _PCG_MULT >> 64
        s_lo, s_hi = inc_lo, inc_hi                                      # state = 0; step: state = 0 * mult + inc
        s_lo, s_hi = _add128(s_lo, s_hi, init_lo, init_hi)               # state += initstate
        s_lo, s_hi = _mul128(s_lo, s_hi, m_lo, m_hi)                     # step
        s_lo, s_hi = _add128(s_lo, s_hi, inc_lo, inc_hi)
    return np.stack([s_lo, s_hi, inc_lo, inc_hi], axis=1)


def _u32_words(value: int) -> list:
    """SeedSequence's coercion of one non-negative int: its little-endian 32-bit words (0 -> [0])."""
    value = int(value)
    if value < 0:
        raise ValueError("seeds must be non-negative")
    out = [value & 0xFFFFFFFF]
    value >>= 32
    while value:
        out.append(value & 0xFFFFFFFF)
        value >>= 32
    return out


def words_from_seeds(seeds) -> np.ndarray:
    """Per-env states for per-env integer seeds (< 2^32 each): env b gets Generator(PCG64(SeedSequence(seeds[b])))."""
    seeds = np.asarray(seeds)
    if seeds.dtype.kind not in "iu":
        raise TypeError("seeds must be integers")
    if seeds.size and seeds.dtype.kind == "i" and int(seeds.min()) < 0:   # (a cast to uint64 would wrap them into other streams)
        raise ValueError("seeds must be non-negative")
    seeds = seeds.astype(np.uint64)
    if seeds.size and int(seeds.max()) >> 32:
        return np.stack([words_from_seed(int(s)) for s in seeds])
    return _pcg64_words_from_entropy([seeds.astype(np.uint32)])


def words_from_seed_and_index(seed: int, global_index) -> np.ndarray:
    """Env with global index g gets Generator(PCG64(SeedSequence([seed, g]))); g == 0 gets SeedSequence(seed), the
    reference's single env (multigrid/base.py:269).  Vectorised over the envs (numpy's own algorithm, restated above)."""
    idx = np.asarray(global_index, dtype=np.uint64).reshape(-1)
    out = np.empty((len(idx), 4), dtype=np.uint64)
    if len(idx) == 0:
        return out
    if int(idx.max()) >> 32:                                             # (indices beyond 2^32: one SeedSequence per env)
        for b, g in enumerate(idx):
            out[b] = words_from_seed(int(seed) if int(g) == 0 else [int(seed), int(g)])
        return out
    sw = [np.full(len(idx), w, dtype=np.uint32) for w in _u32_words(seed)]
    out[:] = _pcg64_words_from_entropy(sw + [idx.astype(np.uint32)])
    zero = np.nonzero(idx == 0)[0]
    if len(zero):
        out[zero] = words_from_seed(int(seed))
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
    idx = np.asarray(global_index, dtype=np.uint64).reshape(-1)
    out = np.zeros((len(idx), 6), dtype=np.uint64)
    if len(idx) == 0:
        return out
    if int(idx.max()) >> 32:
        for b, g in enumerate(idx):
            out[b, :4] = words_from_seed([int(layout_seed), int(g)])
        return out
    sw = [np.full(len(idx), w, dtype=np.uint32) for w in _u32_words(layout_seed)]
    out[:, :4] = _pcg64_words_from_entropy(sw + [idx.astype(np.uint32)])
    return out
