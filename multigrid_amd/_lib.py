"""ctypes binding of libmgx.so (include/mgx.h).  Fails loudly when the HIP library is missing: there is no
CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

from .spec import MgxSpecC

HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB_PATH = os.path.join(HERE, "lib", "libmgx.so")
#: MGX_LIBMGX=<path> loads another build of the library (the profiling tools' -DMGX_DEBUG_KNOBS / experiment builds);
#: bench.py marks every line measured that way `"valid": false`.
LIB_PATH = os.environ.get("MGX_LIBMGX") or PRODUCT_LIB_PATH


def is_product_lib() -> bool:
    return os.path.realpath(LIB_PATH) == os.path.realpath(PRODUCT_LIB_PATH)

ABI_VERSION = 11
OK, ERR_INVALID_ARGUMENT, ERR_UNKNOWN_ACTION, ERR_UNSUPPORTED, ERR_LAUNCH = 0, -1, -2, -3, -4

#: every symbol include/mgx.h declares
EXPORTS = ("mgx_abi_version", "mgx_error_string", "mgx_last_hip_error", "mgx_gen_obs", "mgx_step",
           "mgx_launch_info", "mgx_one_hot", "mgx_full_obs", "mgx_reset_done", "mgx_rollout", "mgx_step_autoreset",
           "mgx_rollout_autoreset", "mgx_gen_obs_one_hot", "mgx_step_one_hot",
           "mgx_reset_generate", "mgx_step_generate", "mgx_pack_grid", "mgx_unpack_grid",
           "mgx_step_ex", "mgx_step_chains", "mgx_sub_shards",
           "mgx_pack_grid_env", "mgx_check_grid", "mgx_pack_grid8_env", "mgx_unpack_grid8", "mgx_shape_key", "mgx_shape_register", "mgx_stage_generate",
           "mgx_persistent_waves", "mgx_step_persistent", "mgx_persistent_post", "mgx_persistent_wait", "mgx_persistent_feed",
           "mgx_rollout_info")


class MgxLaunchInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("envs_per_wavefront", "envs_per_workgroup", "threads_per_workgroup", "workgroups",
                                          "lds_bytes", "slots_per_group", "fixed_shape")]


class MgxRolloutInfo(C.Structure):
    """include/mgx.h: struct MgxRolloutInfo (ABI 11)."""
    _fields_ = [(n, C.c_int32) for n in ("envs_per_slice", "slices", "wavefronts", "threads_per_workgroup", "workgroups", "lds_bytes",
                                          "resident_shape")]


class MgxAutoReset(C.Structure):
    """include/mgx.h: struct MgxAutoReset."""
    _fields_ = [("first_env", C.c_int64), ("pool_size", C.c_int32), ("pool_grid", C.c_void_p),
                ("pool_agents", C.c_void_p), ("pool_aux", C.c_void_p), ("episode", C.c_void_p), ("was_reset", C.c_void_p)]


class MgxGenStage(C.Structure):
    """include/mgx.h: struct MgxGenStage (staged generation of truncation resets)."""
    _fields_ = [("grid", C.c_void_p), ("agents", C.c_void_p), ("aux", C.c_void_p), ("words", C.c_void_p), ("tag", C.c_void_p),
                ("phase", C.c_int32), ("lead", C.c_int32), ("external", C.c_int32), ("candidates", C.c_int32)]


class MgxLayoutGen(C.Structure):
    """include/mgx.h: struct MgxLayoutGen."""
    _fields_ = [("kind", C.c_int32), ("room_size", C.c_int32), ("start_x", C.c_int32), ("start_y", C.c_int32),
                ("start_dir", C.c_int32), ("max_hallway_keys", C.c_int32), ("max_keys_per_room", C.c_int32),
                ("blank", C.c_void_p), ("gen_state", C.c_void_p), ("stage", MgxGenStage)]


class MgxStepArgs(C.Structure):
    """include/mgx.h: struct MgxStepArgs (the general form of the step)."""
    _fields_ = [("grid", C.c_void_p), ("agents", C.c_void_p), ("rng", C.c_void_p), ("step_count", C.c_void_p),
                ("aux", C.c_void_p), ("actions", C.c_void_p), ("hook_order", C.c_void_p),
                ("obs", C.c_void_p), ("dir", C.c_void_p), ("reward", C.c_void_p), ("terminated", C.c_void_p),
                ("truncated", C.c_void_p), ("err", C.c_void_p),
                ("steps", C.c_int32), ("one_hot", C.c_int32),
                ("auto_reset", C.POINTER(MgxAutoReset)), ("generate", C.POINTER(MgxLayoutGen)),
                ("episode", C.c_void_p), ("was_reset", C.c_void_p), ("grid_bad", C.c_void_p)]


class MgxPersistent(C.Structure):
    """include/mgx.h: struct MgxPersistent (persistent stepping)."""
    _fields_ = [("action_granules", C.c_void_p), ("done", C.c_void_p), ("ctrl", C.c_void_p), ("max_steps", C.c_int32),
                ("timeout_ms", C.c_int32)]


PERSIST_CTRL_WORDS = 8

GEN_KINDS = {"empty_fixed": 0, "empty_random": 1, "blockedunlockpickup": 2, "redbluedoors": 3, "lockedhallway": 4, "playground": 5}


class MgxError(RuntimeError):
    def __init__(self, code: int, what: str):
        self.code = code
        super().__init__(f"{what}: {error_string(code)} (code {code}, hip error {lib().mgx_last_hip_error()})")


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `python -m multigrid_amd.build` "
            "(or __graft_entry__.build()). multigrid_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i64 = C.c_void_p, C.c_int64
    L.mgx_abi_version.restype = C.c_int
    L.mgx_error_string.restype = C.c_char_p
    L.mgx_error_string.argtypes = [C.c_int]
    L.mgx_last_hip_error.restype = C.c_int
    L.mgx_gen_obs.restype = C.c_int
    L.mgx_gen_obs.argtypes = [C.POINTER(MgxSpecC), i64, vp, vp, vp, vp, vp]
    L.mgx_step.restype = C.c_int
    L.mgx_step.argtypes = [C.POINTER(MgxSpecC), i64] + [vp] * 13
    L.mgx_rollout.restype = C.c_int
    L.mgx_rollout.argtypes = [C.POINTER(MgxSpecC), i64, C.c_int32] + [vp] * 13
    L.mgx_step_autoreset.restype = C.c_int
    L.mgx_step_autoreset.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxAutoReset)] + [vp] * 13
    L.mgx_rollout_autoreset.restype = C.c_int
    L.mgx_rollout_autoreset.argtypes = [C.POINTER(MgxSpecC), i64, C.c_int32, C.POINTER(MgxAutoReset)] + [vp] * 13
    L.mgx_gen_obs_one_hot.restype = C.c_int
    L.mgx_gen_obs_one_hot.argtypes = [C.POINTER(MgxSpecC), i64, vp, vp, vp, vp, vp]
    L.mgx_step_one_hot.restype = C.c_int
    L.mgx_step_one_hot.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxAutoReset)] + [vp] * 13
    L.mgx_one_hot.restype = C.c_int
    L.mgx_one_hot.argtypes = [vp, i64, C.POINTER(C.c_int32), vp, vp]
    L.mgx_full_obs.restype = C.c_int
    L.mgx_full_obs.argtypes = [C.POINTER(MgxSpecC), i64, vp, vp, vp, vp]
    L.mgx_reset_done.restype = C.c_int
    L.mgx_reset_done.argtypes = [C.POINTER(MgxSpecC), i64, i64, C.c_int32] + [vp] * 10
    L.mgx_reset_generate.restype = C.c_int
    L.mgx_reset_generate.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxLayoutGen)] + [vp] * 8
    L.mgx_stage_generate.restype = C.c_int
    L.mgx_stage_generate.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxLayoutGen), vp, vp, vp]
    L.mgx_step_generate.restype = C.c_int
    L.mgx_step_generate.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxLayoutGen)] + [vp] * 15
    L.mgx_pack_grid.restype = C.c_int
    L.mgx_pack_grid.argtypes = [vp, i64, vp, vp, vp]
    L.mgx_unpack_grid.restype = C.c_int
    L.mgx_unpack_grid.argtypes = [vp, i64, vp, vp]
    L.mgx_pack_grid_env.restype = C.c_int
    L.mgx_pack_grid_env.argtypes = [vp, i64, C.c_int32, C.c_int32, vp, vp, vp]
    L.mgx_pack_grid8_env.restype = C.c_int
    L.mgx_pack_grid8_env.argtypes = [vp, i64, C.c_int32, C.c_int32, vp, vp, vp]
    L.mgx_unpack_grid8.restype = C.c_int
    L.mgx_unpack_grid8.argtypes = [vp, i64, vp, vp]
    L.mgx_check_grid.restype = C.c_int
    L.mgx_check_grid.argtypes = [C.POINTER(MgxSpecC), i64, vp, vp, vp, vp]
    L.mgx_launch_info.restype = C.c_int
    L.mgx_launch_info.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxLaunchInfo)]
    L.mgx_step_ex.restype = C.c_int
    L.mgx_step_ex.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxStepArgs), vp]
    L.mgx_step_chains.restype = C.c_int
    L.mgx_step_chains.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxStepArgs), C.c_int32, C.POINTER(vp), vp]
    L.mgx_sub_shards.restype = C.c_int
    L.mgx_sub_shards.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxStepArgs), C.POINTER(C.c_int32)]
    L.mgx_persistent_waves.restype = C.c_int
    L.mgx_persistent_waves.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxStepArgs), C.POINTER(C.c_int32)]
    L.mgx_step_persistent.restype = C.c_int
    L.mgx_step_persistent.argtypes = [C.POINTER(MgxSpecC), i64, C.POINTER(MgxStepArgs), C.POINTER(MgxPersistent), vp]
    L.mgx_persistent_post.restype = C.c_int
    L.mgx_persistent_post.argtypes = [C.POINTER(MgxSpecC), i64, vp, C.c_uint32, vp, vp]
    L.mgx_persistent_wait.restype = C.c_int
    L.mgx_persistent_wait.argtypes = [vp, C.c_int32, C.c_uint32, vp, C.c_int32, vp]
    L.mgx_persistent_feed.restype = C.c_int
    L.mgx_persistent_feed.argtypes = [C.POINTER(MgxSpecC), i64, vp, C.c_int32, C.POINTER(MgxPersistent), C.c_int32, vp, vp]
    if L.mgx_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {L.mgx_abi_version()} != {ABI_VERSION}; rebuild it")
    _lib = L
    return L


def error_string(code: int) -> str:
    return lib().mgx_error_string(code).decode()


def check(code: int, what: str):
    if code != OK:
        raise MgxError(code, what)


def launch_info(spec, batch: int, roll: bool = False, persistent: bool = False) -> dict:
    """Launch geometry of the plain step, of mgx_rollout* (`roll=True`) or of mgx_step_persistent (`persistent=True`): the latter two
    as MgxRolloutInfo with `envs_per_wavefront` added."""
    sc = spec.to_c()
    if roll or persistent:
        ri = MgxRolloutInfo()
        L = lib()
        L.mgx_rollout_info.restype = C.c_int
        L.mgx_rollout_info.argtypes = [C.POINTER(MgxSpecC), C.c_int64, C.c_int32, C.POINTER(MgxRolloutInfo)]
        check(L.mgx_rollout_info(C.byref(sc), batch, int(bool(persistent)), C.byref(ri)), "mgx_rollout_info")
        out = {n: getattr(ri, n) for n, _ in MgxRolloutInfo._fields_}
        out["envs_per_wavefront"] = ri.envs_per_slice * ri.slices
        return out
    info = MgxLaunchInfo()
    check(lib().mgx_launch_info(C.byref(sc), batch, C.byref(info)), "mgx_launch_info")
    return {n: getattr(info, n) for n, _ in MgxLaunchInfo._fields_}
