"""The five synthetic workloads of BASELINE.json / SURVEY.md section 8(d), defined once for bench.py, the full-size
parity tests and the profiling tools.

    C1  MultiGrid-Empty-8x8-v0, 2 agents, batch 1                   (the reference's own CPU-runnable case)
    C2  MultiGrid-Empty-16x16-v0, 4 agents, view 7, 4096 envs
    C3  MultiGrid-BlockedUnlockPickup-v0 (11x6), 2 agents, 16384 envs, starts drawn from a pool of K = 256 layouts made by
        `layouts.blockedunlockpickup_layout` (the reference's generator restated draw for draw), joint reward + the
        pick-up-the-box hook, auto-reset from the same pool
    C4  C2 at 65536 envs (the north-star configuration)
    C5  custom 64x64 walled grid, goal at (62,62), 16 agents, view 9, 32768 envs, WITH occluders: 5 % of the interior
        cells are walls, plus 8 doors (4 closed, 2 locked, 2 open), 4 keys, 2 balls, 2 boxes per layout; a pool of K = 64
        such layouts from `numpy.random.default_rng(5)`; agent starts / directions are random free interior cells.
        Held on the device as COMPACT one-byte cells (EnvSpec.cell_bytes = 1, include/mgx.h: MgxCell8 -- the format for large
        grids; `make("c5", cell_bytes=2)` gives the same workload on 16-bit cells)

Everything is a pure function of the GLOBAL env index (layout choice, agent starts, PCG64 words), so a shard
[first_env, first_env + batch) of a workload is bit-identical to the same envs inside the whole batch.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from . import layouts, rng as rnglib
from .constants import Color, State, Type
from .spec import EnvSpec

SEED = 1234          # PCG64 words of env g = rnglib.synthetic_words(seed=SEED) at global index g


@dataclasses.dataclass
class Workload:
    name: str                    # "c2", ...
    title: str                   # the BASELINE.json configuration it implements
    spec: EnvSpec
    global_batch: int            # the configuration's batch (all GPUs together)
    grid: np.ndarray             # u8[B,H,W,3]  (this shard)
    agents: np.ndarray           # u8[B,A,8]
    aux: np.ndarray | None       # u8[B,16] for hook envs
    rng: np.ndarray              # u64[B,4]
    pool: tuple                  # (grid u8[K,H,W,3], agents u8[K,A,8], aux u8[K,16] | None): auto-reset layouts
    first_env: int

    @property
    def batch(self) -> int:
        return self.grid.shape[0]

    def make_env(self, device, backend=None, auto_reset: bool = True):
        from .batched import BatchedMultiGridEnv
        env = BatchedMultiGridEnv(self.spec, self.batch, device, first_env=self.first_env, backend=backend)
        env.load_state(self.grid, self.agents, rng=self.rng, aux=self.aux, validate=False)
        if auto_reset:
            env.set_layout_pool(*self.pool)
        return env


GLOBAL_BATCH = {"c1": 1, "c2": 4096, "c3": 16384, "c4": 65536, "c5": 32768}
TITLES = {
    "c1": "MultiGrid-Empty-8x8-v0 agents=2 batch=1 (BASELINE.json configs[0])",
    "c2": "MultiGrid-Empty-16x16-v0 agents=4 view_size=7 batch=4096 (BASELINE.json configs[1])",
    "c3": "MultiGrid-BlockedUnlockPickup-v0 (11x6) agents=2 view_size=7 batch=16384, K=256 layout pool, joint reward + hook "
          "(BASELINE.json configs[2])",
    "c4": "MultiGrid-Empty-16x16-v0 agents=4 view_size=7 batch=65536 (BASELINE.json configs[3])",
    "c5": "custom 64x64 grid agents=16 view_size=9 batch=32768, 5% interior walls + 8 doors/4 keys/2 balls/2 boxes per "
          "layout, K=64 layouts from default_rng(5) (BASELINE.json configs[4])",
}


def spec_of(name: str, cell_bytes: int | None = None) -> EnvSpec:
    """The configuration's EnvSpec.  `cell_bytes`: None = the format the configuration is stepped in (compact cells for the
    64x64 grid of C5, 16-bit cells for the others), or 1 / 2 to force one."""
    if cell_bytes is not None:
        return dataclasses.replace(spec_of(name), cell_bytes=int(cell_bytes))
    if name == "c1":        # multigrid/envs/__init__.py:44, empty.py:145 (max_steps = 4 * size^2)
        return EnvSpec(8, 8, 2, 7, max_steps=4 * 8 * 8)
    if name in ("c2", "c4"):  # multigrid/envs/__init__.py:46
        return EnvSpec(16, 16, 4, 7, max_steps=4 * 16 * 16)
    if name == "c3":        # multigrid/envs/blockedunlockpickup.py:104-136: room_size 6, max_steps 16 * room_size^2, joint
        return EnvSpec(11, 6, 2, 7, max_steps=16 * 6 * 6, joint_reward=True, env_kind="blockedunlockpickup")
    if name == "c5":
        return EnvSpec(64, 64, 16, 9, max_steps=4 * 64 * 64, cell_bytes=1)
    raise ValueError(f"unknown workload {name!r} (c1..c5)")


_POOL_CACHE: dict = {}


def bup_pool(K: int = 256):
    """K BlockedUnlockPickup episode starts: layout k from default_rng(1000 + k) (placement draws) and
    default_rng(2000 + k) (the door row, drawn from env.np_random in the reference: roomgrid.py:106)."""
    key = ("bup", K)
    if key not in _POOL_CACHE:
        gs, ags, auxs = [], [], []
        for k in range(K):
            g, a, target = layouts.blockedunlockpickup_layout(6, 2, np.random.default_rng(1000 + k),
                                                              np.random.default_rng(2000 + k))
            gs.append(g); ags.append(a); auxs.append(layouts.make_aux("blockedunlockpickup", g, target=target))
        _POOL_CACHE[key] = (np.stack(gs), np.stack(ags), np.stack(auxs))
    return _POOL_CACHE[key]


def stress_pool(K: int = 64, size: int = 64, A: int = 16, seed: int = 5, wall_frac: float = 0.05):
    """K layouts of the large-view stress configuration (build-defined; SURVEY 8(d) C5 with its optional occluders)."""
    key = ("stress", K, size, A, seed, wall_frac)
    if key in _POOL_CACHE:
        return _POOL_CACHE[key]
    r = np.random.default_rng(seed)
    gs, ags = [], []
    n_in = (size - 2) * (size - 2)
    for k in range(K):
        g = np.zeros((size, size, 3), dtype=np.uint8)
        g[..., 0] = Type.empty
        g[0, :] = g[-1, :] = g[:, 0] = g[:, -1] = (Type.wall, Color.grey, 0)
        cells = r.permutation(n_in)                       # interior cells in random order, disjoint uses below
        ys, xs = 1 + cells // (size - 2), 1 + cells % (size - 2)
        keep = ~(((xs == size - 2) & (ys == size - 2)) | ((xs == 1) & (ys == 1)))      # the goal and (1,1) stay free
        ys, xs = ys[keep], xs[keep]
        n_wall = int(round(wall_frac * n_in))
        g[ys[:n_wall], xs[:n_wall]] = (Type.wall, Color.grey, 0)
        p = n_wall
        for state, n in ((State.closed, 4), (State.locked, 2), (State.open, 2)):
            for _ in range(n):
                g[ys[p], xs[p]] = (Type.door, int(r.integers(0, 6)), state); p += 1
        for t, n in ((Type.key, 4), (Type.ball, 2), (Type.box, 2)):
            for _ in range(n):
                g[ys[p], xs[p]] = (t, int(r.integers(0, 6)), 0); p += 1
        g[size - 2, size - 2] = (Type.goal, Color.green, 0)
        ag = np.zeros((A, 8), dtype=np.uint8)
        ag[:, 0] = np.arange(A) % 6
        ag[:, 1] = r.integers(0, 4, size=A)
        ag[:, 2], ag[:, 3] = xs[p:p + A], ys[p:p + A]      # distinct free cells
        ag[:, 5] = Type.empty
        gs.append(g); ags.append(ag)
    _POOL_CACHE[key] = (np.stack(gs), np.stack(ags), None)
    return _POOL_CACHE[key]


def _free_cells(g: np.ndarray):
    ys, xs = np.nonzero(g[..., 0] == Type.empty)
    return xs.astype(np.uint8), ys.astype(np.uint8)


def make(name: str, batch: int | None = None, first_env: int = 0, global_batch: int | None = None,
         seed: int = SEED, cell_bytes: int | None = None) -> Workload:
    """The shard [first_env, first_env + batch) of workload `name` at `global_batch` envs (default: the configuration's
    own batch, all of it).  `cell_bytes`: see spec_of."""
    spec = spec_of(name, cell_bytes)
    G = GLOBAL_BATCH[name] if global_batch is None else int(global_batch)
    B = G - first_env if batch is None else int(batch)
    if first_env < 0 or B < 0 or first_env + B > G:
        raise ValueError(f"shard [{first_env}, {first_env + B}) outside the global batch {G}")
    idx = np.arange(first_env, first_env + B)
    aux = None
    if name in ("c1", "c2", "c4"):
        g1, a1 = layouts.empty_layout(spec.width, spec.num_agents)          # empty.py:151-170: agents at (1,1) facing right
        pool = (g1[None], a1[None], None)                                    # its reset() is this one layout
        grid = np.broadcast_to(g1, (B,) + g1.shape).copy()
        agents = np.broadcast_to(a1, (B,) + a1.shape).copy()
    elif name == "c3":
        pool = bup_pool(256)
        k = np.random.default_rng(3).integers(0, 256, size=G)[idx]            # drawn for the global batch, then sliced
        grid, agents, aux = pool[0][k].copy(), pool[1][k].copy(), pool[2][k].copy()
    else:                                                                     # c5
        pool = stress_pool(64, spec.width, spec.num_agents)
        K = pool[0].shape[0]
        r = np.random.default_rng(5)
        k = (np.arange(G) % K)[idx]
        u = r.random((G, spec.num_agents))[idx]                               # agent starts: random free cells of the layout
        d = r.integers(0, 4, size=(G, spec.num_agents))[idx]
        grid = pool[0][k].copy()
        agents = pool[1][k].copy()
        for kk in range(K):
            sel = np.nonzero(k == kk)[0]
            if len(sel) == 0:
                continue
            xs, ys = _free_cells(pool[0][kk])
            c = (u[sel] * len(xs)).astype(np.int64)
            agents[sel, :, 2], agents[sel, :, 3] = xs[c], ys[c]
        agents[..., 1] = d
    words = rnglib.synthetic_words(B, seed, first_env)
    return Workload(name, TITLES[name], spec, G, grid, agents, aux, words, pool, first_env)
