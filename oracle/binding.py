"""ctypes binding for the CPU oracle (oracle/mgx_oracle.c).

TEST INFRASTRUCTURE.  Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under multigrid_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmgx_oracle.so")

KIND = {"empty": 0, "blockedunlockpickup": 1, "redbluedoors": 2, "lockedhallway": 3, "rules": 4}
AUX = 16
ERR_UNKNOWN_ACTION = -2


class MgoSpec(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "num_agents", "view_size", "max_steps", "see_through_walls",
        "allow_agent_overlap", "joint_reward", "success_any", "failure_any", "env_kind")]


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, "mgx_oracle.c"), os.path.join(HERE, "mgx_layout_oracle.c")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-B", "libmgx_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if os.environ.get("MGX_SANITIZE") == "1":           # tests/test_checked_build.py: the oracle under ASan + UBSan
            out = os.path.join(os.environ.get("MGX_SANITIZE_DIR", "/tmp"), "libmgx_oracle_san.so")
            subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-std=c11",
                                   "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", out,
                                   os.path.join(HERE, "mgx_oracle.c"), os.path.join(HERE, "mgx_layout_oracle.c")])
            _lib = C.CDLL(out)
        else:
            _lib = C.CDLL(build())
        _lib.mgo_max_threads.restype = C.c_int
        for name in ("mgo_gen_obs_ref", "mgo_step_ref", "mgo_step_batch", "mgo_gen_obs_batch"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def make_spec(d: dict) -> MgoSpec:
    """d: the spec dict stored in the golden fixtures / EnvSpec.as_dict()."""
    return MgoSpec(
        width=d["width"], height=d["height"], num_agents=d["num_agents"], view_size=d["view_size"],
        max_steps=d["max_steps"], see_through_walls=int(d["see_through_walls"]),
        allow_agent_overlap=int(d["allow_agent_overlap"]), joint_reward=int(d["joint_reward"]),
        success_any=int(d["success_termination_mode"] == "any"),
        failure_any=int(d["failure_termination_mode"] == "any"),
        env_kind=KIND[d["env_kind"]])


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def pcg64_random(state4: np.ndarray, n: int) -> np.ndarray:
    """state4: u64[4] = [state_lo, state_hi, inc_lo, inc_hi]; advanced in place."""
    out = np.empty(n, dtype=np.float64)
    lib().mgo_pcg64_random(_p(state4, C.c_uint64), C.c_int64(n), _p(out, C.c_double))
    return out


def gen_obs_ref(grid_state: np.ndarray, agent_state: np.ndarray, view: int, see_through: bool) -> np.ndarray:
    """Reference-layout single env: grid_state (W,H,3) int64, agent_state (A,9) int64 -> (A,v,v,3) int64."""
    g = np.ascontiguousarray(grid_state, dtype=np.int64)
    a = np.ascontiguousarray(agent_state, dtype=np.int64)
    W, H, _ = g.shape
    A = a.shape[0]
    out = np.empty((A, view, view, 3), dtype=np.int64)
    rc = lib().mgo_gen_obs_ref(_p(g, C.c_int64), _p(a, C.c_int64), W, H, A, view, int(see_through),
                               _p(out, C.c_int64))
    if rc:
        raise ValueError(f"mgo_gen_obs_ref failed: {rc}")
    return out


class RefEnv:
    """Single env in the reference's own array shapes, stepped by the oracle."""

    def __init__(self, spec: dict, grid_state, agent_state, rng_lohi, target=None, step_count=0):
        self.spec_dict = dict(spec)
        self.spec = make_spec(spec)
        self.grid_state = np.ascontiguousarray(grid_state, dtype=np.int64).copy()
        self.agent_state = np.ascontiguousarray(agent_state, dtype=np.int64).copy()
        self.rng = np.ascontiguousarray(rng_lohi, dtype=np.uint64).copy()
        self.step_count = np.array([step_count], dtype=np.int64)
        t = list(target) if target is not None else []
        self.target = np.array(t + [0] * (AUX - len(t)), dtype=np.int64)          # the env subclass' hook state (aux)

    def gen_obs(self):
        return gen_obs_ref(self.grid_state, self.agent_state, self.spec.view_size,
                           bool(self.spec.see_through_walls))

    def step(self, actions, hook_order=None):
        """hook_order: agent indices in the insertion order of the caller's actions dict (None = ascending)."""
        A, v = self.spec.num_agents, self.spec.view_size
        ho = None if hook_order is None else np.ascontiguousarray(hook_order, dtype=np.uint8)
        assert ho is None or ho.shape == (A,)
        act = np.ascontiguousarray(actions, dtype=np.int8)
        obs = np.empty((A, v, v, 3), dtype=np.int64)
        direction = np.empty(A, dtype=np.int64)
        reward = np.empty(A, dtype=np.float64)
        terminated = np.empty(A, dtype=np.uint8)
        truncated = np.zeros(1, dtype=np.uint8)
        order = np.empty(A, dtype=np.int32)
        rc = lib().mgo_step_ref(
            C.byref(self.spec), _p(self.grid_state, C.c_int64), _p(self.agent_state, C.c_int64),
            _p(self.rng, C.c_uint64), _p(self.step_count, C.c_int64), _p(act, C.c_int8),
            _p(self.target, C.c_int64), _p(obs, C.c_int64), _p(direction, C.c_int64), _p(reward, C.c_double),
            _p(terminated, C.c_uint8), _p(truncated, C.c_uint8), _p(order, C.c_int),
            _p(ho, C.c_uint8) if ho is not None else None)
        if rc == ERR_UNKNOWN_ACTION:
            raise ValueError("Unknown action")
        if rc:
            raise RuntimeError(f"mgo_step_ref failed: {rc}")
        return obs, direction, reward, terminated.astype(bool), bool(truncated[0]), order


def step_outputs(spec: dict, B: int):
    """Output buffers of step_batch, to reuse across calls (fresh 38 MB arrays page-fault on every call at 65 536 envs)."""
    sp = make_spec(spec)
    A, v = sp.num_agents, sp.view_size
    return (np.zeros((B, A, v, v, 3), dtype=np.uint8), np.zeros((B, A), dtype=np.uint8), np.zeros((B, A), dtype=np.float64),
            np.zeros((B, A), dtype=np.uint8), np.zeros((B,), dtype=np.uint8))


def step_batch(spec: dict, grid, agents, rng, step_count, actions, target=None, nthreads: int = 1, out=None, hook_order=None):
    """Product-layout batched step on numpy arrays (modified in place; `target` = aux u8[B,16], include/mgx.h).
    Returns (obs, dir, reward, terminated, truncated) -- the arrays of `out` (step_outputs) when given.
    hook_order u8[B,A] | None: the env hooks' visiting order (the caller's dict order; None = ascending index)."""
    sp = make_spec(spec)
    B = grid.shape[0]
    A, v = sp.num_agents, sp.view_size
    for arr in (grid, agents, rng, step_count, actions):
        assert arr.flags.c_contiguous
    if out is not None:
        obs, d, reward, terminated, truncated = out
    else:
        obs = np.empty((B, A, v, v, 3), dtype=np.uint8)
        d = np.empty((B, A), dtype=np.uint8)
        reward = np.empty((B, A), dtype=np.float64)
        terminated = np.empty((B, A), dtype=np.uint8)
        truncated = np.empty((B,), dtype=np.uint8)
    err_env = C.c_int64(-1)
    tgt = _p(target, C.c_uint8) if target is not None else None
    rc = lib().mgo_step_batch(
        C.byref(sp), C.c_int64(B), _p(grid, C.c_uint8), _p(agents, C.c_uint8), _p(rng, C.c_uint64),
        _p(step_count, C.c_int32), _p(actions, C.c_int8), tgt, _p(obs, C.c_uint8), _p(d, C.c_uint8),
        _p(reward, C.c_double), _p(terminated, C.c_uint8), _p(truncated, C.c_uint8), C.byref(err_env),
        int(nthreads), _p(np.ascontiguousarray(hook_order, dtype=np.uint8), C.c_uint8) if hook_order is not None else None)
    if rc == ERR_UNKNOWN_ACTION:
        raise ValueError(f"Unknown action (env {err_env.value})")
    if rc:
        raise RuntimeError(f"mgo_step_batch failed: {rc}")
    return obs, d, reward, terminated, truncated


def gen_obs_batch(spec: dict, grid, agents, nthreads: int = 1):
    sp = make_spec(spec)
    B = grid.shape[0]
    A, v = sp.num_agents, sp.view_size
    obs = np.empty((B, A, v, v, 3), dtype=np.uint8)
    d = np.empty((B, A), dtype=np.uint8)
    rc = lib().mgo_gen_obs_batch(C.byref(sp), C.c_int64(B), _p(grid, C.c_uint8), _p(agents, C.c_uint8),
                                 _p(obs, C.c_uint8), _p(d, C.c_uint8), int(nthreads))
    if rc:
        raise RuntimeError(f"mgo_gen_obs_batch failed: {rc}")
    return obs, d


def max_threads() -> int:
    return int(lib().mgo_max_threads())


def one_hot(cells, dim_sizes=(11, 6, 4)) -> np.ndarray:
    """cells int[..., 3] -> uint8[..., sum(dim_sizes)] (multigrid/wrappers.py:158-190)."""
    x = np.ascontiguousarray(cells, dtype=np.int64)
    n = x.size // 3
    ds = np.array(dim_sizes, dtype=np.int64)
    out = np.empty((n, int(ds.sum())), dtype=np.uint8)
    lib().mgo_one_hot(_p(x, C.c_int64), C.c_int64(n), _p(ds, C.c_int64), _p(out, C.c_uint8))
    return out.reshape(x.shape[:-1] + (int(ds.sum()),))


def full_obs(grid_state, agent_state) -> np.ndarray:
    """Reference-layout single env: (W,H,3), (A,9) -> (W,H,3) int64 (multigrid/wrappers.py:48-58)."""
    g = np.ascontiguousarray(grid_state, dtype=np.int64)
    a = np.ascontiguousarray(agent_state, dtype=np.int64)
    out = np.empty_like(g)
    lib().mgo_full_obs(_p(g, C.c_int64), _p(a, C.c_int64), g.shape[0], g.shape[1], a.shape[0], _p(out, C.c_int64))
    return out


# ---- episode-start generation (oracle/mgx_layout_oracle.c) -----------------------------------------------------------
def gen_words(gen: np.random.Generator) -> np.ndarray:
    """numpy Generator(PCG64) -> u64[5] = [state_lo, state_hi, inc_lo, inc_hi, has_uint32 << 32 | uinteger]."""
    st = gen.bit_generator.state
    s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
    m = (1 << 64) - 1
    return np.array([s & m, s >> 64, inc & m, inc >> 64, (int(st["has_uint32"]) << 32) | int(st["uinteger"])], dtype=np.uint64)


def np_integers(words5: np.ndarray, lo: int, hi: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.int64)
    lib().mgo_np_integers(_p(words5, C.c_uint64), C.c_int64(lo), C.c_int64(hi), C.c_int64(n), _p(out, C.c_int64))
    return out


def bup_layout(room_size: int, A: int, lay_words: np.ndarray, np_words: np.ndarray, blank: np.ndarray):
    """blank: u8[H,W,3] walls only.  Generator words are advanced in place.  Returns (grid, agents u8[A,8], aux u8[16])."""
    grid = np.ascontiguousarray(blank, dtype=np.uint8).copy()
    agents = np.zeros((A, 8), np.uint8); aux = np.zeros(16, np.uint8)
    lib().mgo_bup_layout(room_size, A, _p(lay_words, C.c_uint64), _p(np_words, C.c_uint64), _p(grid, C.c_uint8),
                         _p(agents, C.c_uint8), _p(aux, C.c_uint8))
    return grid, agents, aux


def empty_random_layout(A: int, lay_words: np.ndarray, blank: np.ndarray):
    grid = np.ascontiguousarray(blank, dtype=np.uint8).copy()
    H, W, _ = grid.shape
    agents = np.zeros((A, 8), np.uint8)
    lib().mgo_empty_random_layout(W, H, A, _p(lay_words, C.c_uint64), _p(grid, C.c_uint8), _p(agents, C.c_uint8))
    return grid, agents


def rbd_layout(size: int, A: int, lay_words: np.ndarray, blank: np.ndarray):
    """blank: u8[H,W,3] (outer walls + the middle room's walls).  Returns (grid, agents u8[A,8], aux u8[16])."""
    grid = np.ascontiguousarray(blank, dtype=np.uint8).copy()
    agents = np.zeros((A, 8), np.uint8); aux = np.zeros(16, np.uint8)
    lib().mgo_rbd_layout(size, A, _p(lay_words, C.c_uint64), _p(grid, C.c_uint8), _p(agents, C.c_uint8), _p(aux, C.c_uint8))
    return grid, agents, aux


def np_shuffle(words5: np.ndarray, items) -> list:
    """numpy's Generator.shuffle of a Python list, restated (oracle/mgx_layout_oracle.c lay_shuffle); words advanced in place."""
    x = np.asarray(list(items), dtype=np.int64)
    lib().mgo_np_shuffle(_p(words5, C.c_uint64), _p(x, C.c_int64), C.c_int64(len(x)))
    return [int(v) for v in x]


def lh_layout(num_rooms: int, room_size: int, max_hallway_keys: int, max_keys_per_room: int, A: int, lay_words: np.ndarray,
              blank: np.ndarray):
    """blank: u8[H,W,3] (room walls, hallway opened).  Returns (grid, agents u8[A,8], aux u8[16])."""
    grid = np.ascontiguousarray(blank, dtype=np.uint8).copy()
    agents = np.zeros((A, 8), np.uint8); aux = np.zeros(16, np.uint8)
    lib().mgo_lh_layout(num_rooms, room_size, max_hallway_keys, max_keys_per_room, A, _p(lay_words, C.c_uint64),
                        _p(grid, C.c_uint8), _p(agents, C.c_uint8), _p(aux, C.c_uint8))
    return grid, agents, aux


def playground_layout(room_size: int, num_rows: int, num_cols: int, A: int, lay_words: np.ndarray, np_words: np.ndarray,
                      blank: np.ndarray):
    grid = np.ascontiguousarray(blank, dtype=np.uint8).copy()
    agents = np.zeros((A, 8), np.uint8)
    lib().mgo_playground_layout(room_size, num_rows, num_cols, A, _p(lay_words, C.c_uint64), _p(np_words, C.c_uint64),
                                _p(grid, C.c_uint8), _p(agents, C.c_uint8))
    return grid, agents
