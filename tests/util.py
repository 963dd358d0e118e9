"""Shared helpers for the test-suite (test infrastructure; may import oracle/)."""
from __future__ import annotations

import glob
import json
import os

import numpy as np
import torch

from multigrid_amd import layouts
from multigrid_amd.spec import EnvSpec
from oracle import binding as ob

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_ALL = sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
#: rollout fixtures (state + actions + per-step outputs) and reset/layout fixtures, both written by oracle/gen_golden.py
GOLDEN = [p for p in _ALL if not os.path.basename(p).startswith(("layout_", "wrappers_", "custom_", "customsteps_"))]
GOLDEN_IDS = [os.path.basename(p)[:-4] for p in GOLDEN]
LAYOUT_GOLDEN = [p for p in _ALL if os.path.basename(p).startswith("layout_")]
LAYOUT_IDS = [os.path.basename(p)[:-4] for p in LAYOUT_GOLDEN]
LAYOUT_ENV_IDS = {"layout_bup_a2": "MultiGrid-BlockedUnlockPickup-v0", "layout_bup_a3": "MultiGrid-BlockedUnlockPickup-v0",
                  "layout_emptyrandom6_a3": "MultiGrid-Empty-Random-6x6-v0", "layout_empty8_a2": "MultiGrid-Empty-8x8-v0",
                  "layout_rbd8_a2": "MultiGrid-RedBlueDoors-8x8-v0", "layout_lh4_a2": "MultiGrid-LockedHallway-4Rooms-v0",
                  "layout_playground_a2": "MultiGrid-Playground-v0"}
#: reset sequences of the user-defined envs of tests/custom_envs.py, recorded over the reference (oracle/gen_golden.py: record_custom)
CUSTOM_GOLDEN = [p for p in _ALL if os.path.basename(p).startswith("custom_")]
CUSTOM_IDS = [os.path.basename(p)[:-4] for p in CUSTOM_GOLDEN]
#: whole episodes of user-defined envs (boxes that hold things, `step` overrides with on_success / on_failure) over the reference
#: (oracle/gen_golden.py: record_custom_steps)
CUSTOM_STEPS_GOLDEN = [p for p in _ALL if os.path.basename(p).startswith("customsteps_")]
CUSTOM_STEPS_IDS = [os.path.basename(p)[:-4] for p in CUSTOM_STEPS_GOLDEN]
WRAPPER_GOLDEN = [p for p in _ALL if os.path.basename(p).startswith("wrappers_")]
WRAPPER_IDS = [os.path.basename(p)[:-4] for p in WRAPPER_GOLDEN]


def load_golden(path):
    z = np.load(path)
    d = json.loads(str(z["spec_json"]))
    return z, d, EnvSpec.from_dict(d)


def rng_words_lohi(words_hilo) -> np.ndarray:
    """golden order [state_hi, state_lo, inc_hi, inc_lo] -> product order [state_lo, state_hi, inc_lo, inc_hi]"""
    hs, ls, hi, li = (int(w) for w in words_hilo)
    return np.array([ls, hs, li, hi], dtype=np.uint64)


def golden_aux(d: dict) -> np.ndarray:
    """The env subclass' hook state for a golden fixture, as the product's aux u8[16] (include/mgx.h)."""
    a = np.zeros(16, dtype=np.uint8)
    kind = d.get("env_kind", "empty")
    if kind == "blockedunlockpickup":
        a[:3] = d["target"]
    elif kind == "redbluedoors":
        a[0:2] = d["blue_door"]; a[2:4] = d["red_door"]
    elif kind == "lockedhallway":
        doors = d["doors"]
        if len(doors) <= 6:
            a[0] = len(doors)
            for i, (x, y) in enumerate(doors):
                a[2 + 2 * i], a[3 + 2 * i] = x, y
        else:                                          # geometric format (include/mgx.h): room_size, len(self.rooms)
            a[0] = 0x80 | len(doors)
            a[3] = min(x for x, _ in doors) + 1
            a[4] = d["num_room_keys"]
    return a


golden_target = golden_aux


def random_state(spec: EnvSpec, B: int, seed: int, density: float = 0.25, terminated_p: float = 0.05,
                 carry_p: float = 0.3, box_contents_p: float = 0.0):
    """Random walled grids with every object type, agents on overlappable cells (possibly stacked),
    some carrying, some already terminated.  Product layout.  box_contents_p: the share of the boxes -- lying around and carried
    -- that hold something (include/mgx.h "BOX CONTENTS": the state byte's upper bits)."""
    r = np.random.default_rng(seed)
    H, W, A = spec.height, spec.width, spec.num_agents
    grid = np.zeros((B, H, W, 3), dtype=np.uint8)
    grid[..., 0] = 1
    kinds = np.array([[9, 0, 0], [3, 2, 0], [5, 1, 0], [6, 3, 0], [7, 4, 0], [2, 5, 0], [4, 0, 1], [4, 1, 2],
                      [4, 2, 0], [8, 1, 0]], dtype=np.uint8)
    pick = r.integers(0, len(kinds), size=(B, H, W))
    cells = kinds[pick]
    cells[..., 1] = np.where(np.isin(cells[..., 0], (2, 8, 9)), cells[..., 1], r.integers(0, 6, size=(B, H, W)))
    put = r.random((B, H, W)) < density
    grid[put] = cells[put]
    grid[:, 0, :] = (2, 5, 0); grid[:, -1, :] = (2, 5, 0); grid[:, :, 0] = (2, 5, 0); grid[:, :, -1] = (2, 5, 0)
    agents = np.zeros((B, A, 8), dtype=np.uint8)
    agents[..., 0] = np.arange(A) % 6
    agents[..., 1] = r.integers(0, 4, size=(B, A))
    t, s = grid[..., 0], grid[..., 2]
    ok = (t == 1) | (t == 3) | (t == 8) | (t == 9) | ((t == 4) & (s == 0))
    ok[:, 0, :] = ok[:, -1, :] = False
    ok[:, :, 0] = ok[:, :, -1] = False
    for b in range(B):
        ys, xs = np.nonzero(ok[b])
        if len(xs) == 0:
            grid[b, 1, 1] = (1, 0, 0)
            ys, xs = np.array([1]), np.array([1])
        k = r.integers(0, len(xs), size=A)
        agents[b, :, 2], agents[b, :, 3] = xs[k], ys[k]
    agents[..., 4] = r.random((B, A)) < terminated_p
    carry = kinds[r.integers(2, 5, size=(B, A))]
    carry[..., 1] = r.integers(0, 6, size=(B, A))
    has = r.random((B, A)) < carry_p
    agents[..., 5:8] = np.where(has[..., None], carry, np.array([1, 0, 0], dtype=np.uint8))
    if box_contents_p > 0:
        rc = np.random.default_rng(seed + 2)
        for arr in (grid, agents[..., 5:8]):
            is_box = arr[..., 0] == 7
            code = (rc.integers(1, 8, size=is_box.shape) | (rc.integers(0, 6, size=is_box.shape) << 3)).astype(np.uint8)
            fill = is_box & (rc.random(is_box.shape) < box_contents_p)
            arr[..., 2] = np.where(fill, arr[..., 2] | (code << 2), arr[..., 2])
    rng = np.random.default_rng(seed + 1).integers(0, 2 ** 63, size=(B, 4), dtype=np.int64).astype(np.uint64)
    rng[:, 2] |= np.uint64(1)
    step_count = r.integers(0, max(1, spec.max_steps), size=B).astype(np.int32)
    target = np.zeros((B, 16), dtype=np.uint8)               # aux (include/mgx.h); BlockedUnlockPickup: the target box
    target[:, 0] = 7
    target[:, 1] = r.integers(0, 6, size=B)
    return dict(grid=grid, agents=agents, rng=rng, step_count=step_count, target=target)


def dev_cells(grid3, device):
    """(type, color, state) bytes u8[...,3] (numpy) -> the device's packed cells (int16 tensor) through torch.ops.mgx.pack_grid
    (the kernel is checked against layouts.pack_cells on the way)."""
    import multigrid_amd.ops  # noqa: F401  (registers the ops)
    cells, bad = torch.ops.mgx.pack_grid(torch.from_numpy(np.ascontiguousarray(grid3)).to(device))
    assert int(bad[0]) == 0
    assert np.array_equal(cells.cpu().numpy().view(np.uint16), layouts.pack_cells(grid3))
    return cells


def grid3(cells) -> np.ndarray:
    """packed cells (int16 tensor on the device) -> (type, color, state) bytes (numpy) through torch.ops.mgx.unpack_grid"""
    import multigrid_amd.ops  # noqa: F401
    out = torch.ops.mgx.unpack_grid(cells).cpu().numpy()
    assert np.array_equal(layouts.pack_cells(out).view(np.int16), cells.cpu().numpy()), "opaque bits out of date"
    return out


def random_actions(B, A, seed, p_missing=0.05):
    r = np.random.default_rng(seed)
    a = r.integers(0, 7, size=(B, A)).astype(np.int8)
    a[r.random((B, A)) < p_missing] = -1
    return a


class OracleBackend:
    """Test-only launcher with HipBackend's interface, computing on the CPU oracle.  Lets the `not gpu`
    suite exercise the host logic (BatchedMultiGridEnv, the dict API, sharding).  Never used by product code."""

    name = "oracle"

    def __init__(self, spec: EnvSpec, nthreads: int = 1):
        self.spec, self.d, self.nthreads = spec, spec.as_dict(), nthreads

    # `grid` arrives as BatchedMultiGridEnv holds it: packed cells (int16 MgxCell bit patterns, or the compact MgxCell8 bytes of
    # a spec with cell_bytes == 1); the oracle works on the reference's (type, color, state) triples
    def _g3(self, grid):
        return layouts.unpack_cells_for(self.spec, grid.numpy())

    def _store(self, grid, g3):
        grid.copy_(torch.from_numpy(layouts.pack_cells_for(self.spec, g3)))

    def gen_obs(self, B, grid, agents, obs, dirs, one_hot: bool = False):
        assert not one_hot
        o, d = ob.gen_obs_batch(self.d, self._g3(grid), agents.numpy(), self.nthreads)
        obs.copy_(torch.from_numpy(o))
        if dirs is not None:
            dirs.copy_(torch.from_numpy(d))

    def step(self, B, grid, agents, rng, step_count, actions, target, err, obs, dirs, reward, terminated, truncated,
             hook_order=None, one_hot: bool = False, auto_reset=None):
        if auto_reset is not None:                 # the fused form == reset_done, then the step (include/mgx.h: mgx_step_autoreset)
            first_env, pool, episode, was_reset = auto_reset
            self.reset_done(B, first_env, pool, grid, agents, step_count, target, episode, was_reset)
        g3 = self._g3(grid)
        try:
            o, d, r, te, tr = ob.step_batch(
                self.d, g3, agents.numpy(), rng.numpy().view(np.uint64), step_count.numpy(),
                actions.numpy(), target.numpy() if target is not None else None, self.nthreads,
                hook_order=None if hook_order is None else hook_order.numpy())
        except ValueError:
            err[0] += 1
            err[1] = 0
            return
        self._store(grid, g3)
        obs.copy_(torch.from_numpy(ob.one_hot(o) if one_hot else o)); dirs.copy_(torch.from_numpy(d)); reward.copy_(torch.from_numpy(r))
        terminated.copy_(torch.from_numpy(te)); truncated.copy_(torch.from_numpy(tr))

    def rollout(self, B, T, grid, agents, rng, step_count, actions, target, err, obs, dirs, reward, terminated,
                truncated):
        for t in range(T):
            self.step(B, grid, agents, rng, step_count, actions[t], target, err, obs[t], dirs[t], reward[t],
                      terminated[t], truncated[t])

    def one_hot(self, cells, out):
        out.copy_(torch.from_numpy(ob.one_hot(cells.numpy())))

    def full_obs(self, B, grid, agents, out):
        g, a = self._g3(grid), agents.numpy()
        for b in range(B):
            out[b] = torch.from_numpy(ob.full_obs(layouts.grid_from_product(g[b]), layouts.unpack_agents(a[b])).astype(np.uint8))

    def reset_done(self, B, first_env, pool, grid, agents, step_count, target, episode, was_reset):       # (cells copied as they are)
        pg, pa, pt = pool
        K = pg.shape[0]
        for b in range(B):
            done = bool((agents[b, :, 4] != 0).all()) or int(step_count[b]) >= self.spec.max_steps
            was_reset[b] = int(done)
            if done:
                k = (first_env + b + int(episode[b]) * 7919) % K
                grid[b] = pg[k]; agents[b] = pa[k]
                if pt is not None:
                    target[b] = pt[k]          # aux
                step_count[b] = 0
                episode[b] += 1

    def reset_generate(self, B, gen, grid, agents, rng, step_count, aux, episode, was_reset):
        """mgx_reset_generate on the layout oracle (oracle/mgx_layout_oracle.c), env by env."""
        blank = layouts.unpack_cells(gen["blank"].numpy())
        gs = gen["gen_state"].numpy().view(np.uint64)
        r = rng.numpy().view(np.uint64)
        A = self.spec.num_agents
        for b in range(B):
            done = bool((agents[b, :, 4] != 0).all()) or int(step_count[b]) >= self.spec.max_steps
            was_reset[b] = int(done)
            if not done:
                continue
            lay = gs[b, :5].copy()
            npw = np.concatenate([r[b], gs[b, 5:6]])
            if gen["kind"] == "blockedunlockpickup":
                g, a, x = ob.bup_layout(gen["room_size"], A, lay, npw, blank)
                aux[b] = torch.from_numpy(x)
            elif gen["kind"] == "lockedhallway":
                rs = gen["room_size"]
                g, a, x = ob.lh_layout(2 * ((self.spec.height - 1) // (rs - 1)), rs, gen["max_hallway_keys"], gen["max_keys_per_room"],
                                       A, lay, blank)
                aux[b] = torch.from_numpy(x)
            elif gen["kind"] == "playground":
                rs = gen["room_size"]
                g, a = ob.playground_layout(rs, (self.spec.height - 1) // (rs - 1), (self.spec.width - 1) // (rs - 1), A, lay, npw, blank)
            elif gen["kind"] == "redbluedoors":
                g, a, x = ob.rbd_layout(self.spec.height, A, lay, blank)
                aux[b] = torch.from_numpy(x)
            elif gen["kind"] == "empty_random":
                g, a = ob.empty_random_layout(A, lay, blank)
            else:
                g = blank.copy(); a = np.zeros((A, 8), np.uint8)
                a[:, 0] = np.arange(A) % 6; a[:, 1] = gen["start"][2]; a[:, 2] = gen["start"][0]; a[:, 3] = gen["start"][1]; a[:, 5] = 1
            self._store(grid[b], g); agents[b] = torch.from_numpy(a)
            gs[b, :5] = lay; gs[b, 5] = npw[4]; r[b] = npw[:4]
            step_count[b] = 0
            episode[b] += 1

    def launch_info(self, B):
        return {}
