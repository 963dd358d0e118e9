"""ctypes access to tests/hostshim/hostshim.cpp (test infrastructure; see the .cpp header)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libmgx_hostshim.so")
SRC = os.path.join(HERE, "hostshim.cpp")
RULES = os.path.join(os.path.dirname(os.path.dirname(HERE)), "multigrid_amd", "csrc", "mgx_rules.h")

_lib = None


def lib():
    global _lib
    if _lib is None:
        if os.environ.get("MGX_SANITIZE") == "1":           # tests/test_checked_build.py: ASan + UBSan build, kept elsewhere
            out = os.path.join(os.environ.get("MGX_SANITIZE_DIR", "/tmp"), "libmgx_hostshim_san.so")
            subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
                                   "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", out, SRC])
            _lib = C.CDLL(out)
            return _lib
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(RULES)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall",
                                   "-o", LIB, SRC])
        _lib = C.CDLL(LIB)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _layouts():
    from multigrid_amd import layouts
    return layouts


def step_env(spec, tile, rows8, act, rng4, step_count, target, force_serial=False, hook_order=None):
    """tile u8[H,W,3], rows8 u8[A,8], act i8[A], rng4 u64[4], target = aux u8[16]; all updated in place.
    Returns dict(obs, reward, terminated, truncated, order, rc, n_dirty)."""
    sc = spec.to_c()
    A, v = spec.num_agents, spec.view_size
    # the rules work on packed cells (include/mgx.h MgxCell, or MgxCell8 for spec.cell_bytes == 1); the tests speak (type, color,
    # state) bytes
    L = _layouts()
    tile3, tile = tile, np.ascontiguousarray(L.pack_cells_for(spec, tile))
    over = np.empty_like(tile)
    rew = np.empty(A, np.float64); term = np.empty(A, np.uint8); trunc = np.zeros(1, np.uint8)
    order = np.empty(A, np.uint8); nd = C.c_int32(0); scnt = C.c_int32(int(step_count))
    rows = rows8.view(np.uint64).reshape(A)
    rc = lib().shim_step_env(C.byref(sc), _p(tile, C.c_uint8), _p(over, C.c_uint8), _p(rows, C.c_uint64),
                             _p(act, C.c_int8), _p(rng4, C.c_uint64), C.byref(scnt), _p(target, C.c_uint8),
                             _p(rew, C.c_double), _p(term, C.c_uint8), _p(trunc, C.c_uint8), _p(order, C.c_uint8),
                             C.byref(nd), int(force_serial),
                             _p(np.ascontiguousarray(hook_order, dtype=np.uint8), C.c_uint8) if hook_order is not None else None)
    obs = np.empty((A, v, v, 3), np.uint8)
    assert lib().shim_obs_env(C.byref(sc), _p(over, C.c_uint8), _p(rows, C.c_uint64), _p(obs, C.c_uint8)) == 0
    assert np.array_equal(L.pack_cells_for(spec, L.unpack_cells_for(spec, tile)), tile), "opaque bits out of date"
    tile3[...] = L.unpack_cells_for(spec, tile)
    return dict(obs=obs, reward=rew, terminated=term, truncated=int(trunc[0]), order=order, rc=rc,
                n_dirty=nd.value, step_count=scnt.value, serial=nd.value < 0)


def obs_env(spec, tile, rows8):
    sc = spec.to_c()
    A, v = spec.num_agents, spec.view_size
    over = np.ascontiguousarray(_layouts().pack_cells_for(spec, tile))
    rows = np.ascontiguousarray(rows8).view(np.uint64).reshape(A)
    lib().shim_overlay(C.byref(sc), _p(over, C.c_uint8), _p(rows, C.c_uint64))
    obs = np.empty((A, v, v, 3), np.uint8)
    assert lib().shim_obs_env(C.byref(sc), _p(over, C.c_uint8), _p(rows, C.c_uint64), _p(obs, C.c_uint8)) == 0
    return obs
