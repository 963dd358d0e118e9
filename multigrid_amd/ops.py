"""PyTorch-ROCm custom ops over the C ABI of libmgx.so -- compiled (csrc/mgx_torch.cpp, lib/libmgx_torch.so), loaded here.

    torch.ops.mgx.gen_obs(grid, agents, spec) -> (obs, dir)
    torch.ops.mgx.step(grid!, agents!, rng!, step_count!, actions, aux!?, err!, spec)
                                               -> (obs, dir, reward, terminated, truncated)
    torch.ops.mgx.step_ordered(grid!, agents!, rng!, step_count!, actions, hook_order, aux!?, err!, spec)   (hook_order u8[B,A]: the
                                               env hooks' visiting order = the caller's dict order, include/mgx.h)
    torch.ops.mgx.step_autoreset(grid!, agents!, rng!, step_count!, actions, aux!?, err!, pool_grid, pool_agents,
                                 pool_aux?, episode!, first_env, spec)
                                               -> (obs, dir, reward, terminated, truncated, was_reset)
    torch.ops.mgx.rollout(grid!, agents!, rng!, step_count!, actions[T,B,A], aux!?, err!, spec)
                                               -> (obs[T,...], dir, reward, terminated, truncated)
    torch.ops.mgx.one_hot(cells, dim_sizes) -> one_hot        torch.ops.mgx.full_obs(grid, agents, spec) -> full
    torch.ops.mgx.pack_grid(cells3 u8[...,3]) -> (grid i16[...], bad i32[1])     torch.ops.mgx.unpack_grid(grid) -> cells3

`grid` / `pool_grid` are PACKED cells, int16 tensors [B,H,W] holding the MgxCell bit patterns of include/mgx.h (type |
color << 8 | state << 12 | opaque << 15); pack_grid / unpack_grid convert from / to (type, color, state) bytes on the device.
The same ops also accept `grid` / `pool_grid` as those bytes, uint8 [B,H,W,3] (packed on the way in and, for the mutating ops,
unpacked back into the caller's tensor on the way out: two extra streaming kernels per call, no host synchronisation; cells the
packed format cannot hold are stored truncated -- pack_grid reports their count).

`spec` is the 11-int list of `struct MgxSpec` (include/mgx.h).  Only the CUDA (= HIP on ROCm) dispatch key is
registered: calling the ops with CPU tensors raises NotImplementedError from the dispatcher -- there is no
CPU implementation to fall back to.  Kernels are enqueued on torch's current HIP stream for the tensors'
device; nothing synchronises.

`HipBackend` is the allocation-free form used by BatchedMultiGridEnv (pre-bound output buffers).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from .spec import EnvSpec, MgxSpecC

_SPEC_FIELDS = [n for n, _ in MgxSpecC._fields_]


def spec_to_ints(spec: EnvSpec) -> list[int]:
    c = spec.to_c()
    return [int(getattr(c, n)) for n in _SPEC_FIELDS]


# torch's current HIP stream of a device as a raw handle, without building a torch.cuda.Stream object (2 us -> 0.2 us per call)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device


def _stream(device) -> int:
    if _raw_stream is not None:
        index = device.index
        return _raw_stream(index if index is not None else _cur_device())
    return torch.cuda.current_stream(device).cuda_stream


def _gen_obs_into(sc: MgxSpecC, B, grid, agents, obs, dirs, one_hot: bool = False):
    fn = _lib.lib().mgx_gen_obs_one_hot if one_hot else _lib.lib().mgx_gen_obs
    with torch.cuda.device(grid.device):
        rc = fn(C.byref(sc), B, grid.data_ptr(), agents.data_ptr(), obs.data_ptr(),
                dirs.data_ptr() if dirs is not None else None, _stream(grid.device))
    _lib.check(rc, "mgx_gen_obs_one_hot" if one_hot else "mgx_gen_obs")


def _step_into(sc: MgxSpecC, B, grid, agents, rng, step_count, actions, target, err, obs, dirs, reward,
               terminated, truncated):
    with torch.cuda.device(grid.device):
        rc = _lib.lib().mgx_step(
            C.byref(sc), B, grid.data_ptr(), agents.data_ptr(), rng.data_ptr() if rng is not None else None,
            step_count.data_ptr(), actions.data_ptr(), target.data_ptr() if target is not None else None,
            obs.data_ptr(), dirs.data_ptr(), reward.data_ptr(), terminated.data_ptr(), truncated.data_ptr(),
            err.data_ptr() if err is not None else None, _stream(grid.device))
    _lib.check(rc, "mgx_step")


def _ptr(t):
    return t.data_ptr() if t is not None else None


ONE_HOT_DIMS = (11, 6, 4)      # len(Type), len(Color), max(len(State), len(Direction))  (multigrid/wrappers.py:139-140)


def _one_hot_into(cells, out, dim_sizes=ONE_HOT_DIMS):
    ds = (C.c_int32 * 3)(*dim_sizes)
    with torch.cuda.device(cells.device):
        rc = _lib.lib().mgx_one_hot(cells.data_ptr(), cells.numel() // 3, ds, out.data_ptr(), _stream(cells.device))
    _lib.check(rc, "mgx_one_hot")


def _full_obs_into(sc: MgxSpecC, B, grid, agents, out):
    with torch.cuda.device(grid.device):
        rc = _lib.lib().mgx_full_obs(C.byref(sc), B, grid.data_ptr(), agents.data_ptr(), out.data_ptr(),
                                     _stream(grid.device))
    _lib.check(rc, "mgx_full_obs")


# The operators themselves are COMPILED: csrc/mgx_torch.cpp -> lib/libmgx_torch.so, TORCH_LIBRARY(mgx) + TORCH_LIBRARY_IMPL(mgx,
# CUDA), in the dispatcher without any Python (a C++ program that links libtorch loads the same file).  Loading it here is what
# makes `torch.ops.mgx.*` exist; a missing library is an ImportError, never a Python fallback.
TORCH_LIB_PATH = os.path.join(os.path.dirname(_lib.PRODUCT_LIB_PATH), "libmgx_torch.so")


def _load_compiled_ops():
    if not os.path.exists(TORCH_LIB_PATH):
        raise ImportError(f"{TORCH_LIB_PATH} is missing: the compiled operator library has not been built. Run "
                          "`python -m multigrid_amd.build --torch` (or __graft_entry__.build()).")
    _lib.lib()                                       # libmgx.so first (the operator library links against it)
    torch.ops.load_library(TORCH_LIB_PATH)
    if int(torch.ops.mgx.abi_version()) != _lib.ABI_VERSION:
        raise ImportError(f"{TORCH_LIB_PATH} was built against ABI {int(torch.ops.mgx.abi_version())}, libmgx.so is {_lib.ABI_VERSION}")


_load_compiled_ops()


class HipBackend:
    """Pre-bound launcher: the state and output tensors of one BatchedMultiGridEnv, validated once."""

    name = "hip"

    def __init__(self, spec: EnvSpec, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError(
                f"multigrid_amd runs on MI355X only: device must be a HIP ('cuda') device, got '{device}'. "
                "There is no CPU fallback.")
        if not torch.cuda.is_available():
            raise RuntimeError("multigrid_amd: no HIP device is visible (torch.cuda.is_available() is False)")
        _lib.lib()  # raises ImportError if the extension is missing
        self.spec = spec
        self.sc = spec.to_c()
        self.device = device

    def gen_obs(self, B, grid, agents, obs, dirs, one_hot: bool = False):
        _gen_obs_into(self.sc, B, grid, agents, obs, dirs, one_hot)

    @staticmethod
    def _auto_reset_struct(auto_reset):
        """auto_reset = (first_env, (pool_grid, pool_agents, pool_aux | None), episode, was_reset | None)"""
        first_env, (pg, pa, pt), episode, was_reset = auto_reset
        return _lib.MgxAutoReset(first_env, pg.shape[0], pg.data_ptr(), pa.data_ptr(),
                                 pt.data_ptr() if pt is not None else None, episode.data_ptr(),
                                 was_reset.data_ptr() if was_reset is not None else None)

    def step(self, B, grid, agents, rng, step_count, actions, target, err, obs, dirs, reward, terminated, truncated,
             auto_reset=None, one_hot: bool = False):
        """`one_hot`: `obs` is the u8[B,A,v,v,21] one-hot buffer (mgx_step_one_hot)."""
        if auto_reset is None and not one_hot:
            _step_into(self.sc, B, grid, agents, rng, step_count, actions, target, err, obs, dirs, reward,
                       terminated, truncated)
            return
        ar = self._auto_reset_struct(auto_reset) if auto_reset is not None else None
        fn = _lib.lib().mgx_step_one_hot if one_hot else _lib.lib().mgx_step_autoreset
        with torch.cuda.device(grid.device):
            rc = fn(
                C.byref(self.sc), B, C.byref(ar) if ar is not None else None, grid.data_ptr(), agents.data_ptr(),
                rng.data_ptr() if rng is not None else None, step_count.data_ptr(), actions.data_ptr(),
                target.data_ptr() if target is not None else None, obs.data_ptr(), dirs.data_ptr(), reward.data_ptr(),
                terminated.data_ptr(), truncated.data_ptr(), err.data_ptr() if err is not None else None,
                _stream(grid.device))
        _lib.check(rc, "mgx_step_one_hot" if one_hot else "mgx_step_autoreset")

    @staticmethod
    def _layout_gen_struct(gen):
        sx, sy, sd = gen.get("start", (0, 0, 0))
        g = _lib.MgxLayoutGen(_lib.GEN_KINDS[gen["kind"]], int(gen.get("room_size", 0)), int(sx), int(sy), int(sd),
                              int(gen.get("max_hallway_keys", 1)), int(gen.get("max_keys_per_room", 2)),
                              gen["blank"].data_ptr(), gen["gen_state"].data_ptr())
        st = gen.get("stage")
        if st is not None:                      # staged generation of truncation resets (include/mgx.h: MgxGenStage)
            g.stage = _lib.MgxGenStage(st["grid"].data_ptr(), st["agents"].data_ptr(), _ptr(st.get("aux")), st["words"].data_ptr(),
                                       st["tag"].data_ptr(), int(st["phase"][0]), int(st.get("lead", 2)), int(st.get("external") or 0),
                                       int(st.get("candidates") or 0))
        return g

    def step_args(self, grid, agents, rng, step_count, target, err, obs, dirs, reward, terminated, truncated,
                  auto_reset=None, one_hot: bool = False, generate=None):
        """The MgxStepArgs of one env's step (include/mgx.h) with every pointer but `actions` / `hook_order` resolved.
        Returns (args, keep): `keep` holds the structs `args` points into."""
        ar = self._auto_reset_struct(auto_reset) if auto_reset is not None else None
        sa = _lib.MgxStepArgs()
        sa.grid, sa.agents, sa.rng, sa.step_count = grid.data_ptr(), agents.data_ptr(), _ptr(rng), step_count.data_ptr()
        sa.aux, sa.err = _ptr(target), _ptr(err)
        sa.obs, sa.dir, sa.reward = obs.data_ptr(), _ptr(dirs), reward.data_ptr()
        sa.terminated, sa.truncated = terminated.data_ptr(), truncated.data_ptr()
        sa.steps, sa.one_hot = 1, int(bool(one_hot))
        if self.spec.cell_bytes == 3:               # byte grids: what the kernel's own packing counts (include/mgx.h: grid_bad)
            sa.grid_bad = self.grid_bad().data_ptr()
        gen = None
        if ar is not None:
            sa.auto_reset = C.pointer(ar)
        if generate is not None:                    # generate = (gen dict, episode, was_reset): mgx_step_generate
            gd, episode, was_reset = generate
            gen = self._layout_gen_struct(gd)
            sa.generate = C.pointer(gen)
            sa.episode, sa.was_reset = episode.data_ptr(), _ptr(was_reset)
        return sa, (ar, gen, self.sc)

    def grid_bad(self):
        """i32[2] on the device: cell values the packed format cannot hold / outer-ring cells that are not WALL, as counted by the
        step kernel while it packs a byte grid (MgxSpec.cell_bytes = 3); BatchedMultiGridEnv.check_errors() reads it."""
        if getattr(self, "_grid_bad", None) is None:
            self._grid_bad = torch.zeros(2, dtype=torch.int32, device=self.device)
        return self._grid_bad

    def bind_step(self, B, grid, agents, rng, step_count, target, err, obs, dirs, reward, terminated, truncated,
                  auto_reset=None, one_hot: bool = False, generate=None):
        """Pre-bound launcher for a policy-in-the-loop caller: every pointer except `actions` (and the optional `hook_order`)
        is resolved once into one MgxStepArgs, so a call costs one 4-argument ctypes transition (mgx_step_ex) + the kernel
        launch.  The tensors must stay alive and in place (they are the env's own buffers).  Returns f(actions, hook_order=None)
        enqueuing one step on torch's current stream."""
        fn = _lib.lib().mgx_step_ex
        sa, keep = self.step_args(grid, agents, rng, step_count, target, err, obs, dirs, reward, terminated, truncated,
                                  auto_reset, one_hot, generate)
        spec_ref, args_ref = C.byref(self.sc), C.byref(sa)
        dev, index = grid.device, grid.device.index
        device_ctx = torch.cuda.device

        gen_c = keep[1]
        stage = generate[0].get("stage") if generate is not None else None
        phase = stage["phase"] if stage is not None else None
        # staging with the generator BESIDE the steps (MgxGenStage.external): every `every` steps the pending snapshot requests are
        # served by a launch of their own on the env's generator stream, ordered behind the step that took them
        side = stage is not None and bool(stage.get("external"))
        if side:
            gen_stream, every = stage["stream"], max(1, int(stage.get("lead", 2)) // 2)
            stage_fn, gen_ref = _lib.lib().mgx_stage_generate, C.byref(gen_c)
            rng_ptr, episode_ptr = rng.data_ptr(), generate[1].data_ptr()

        def step(actions, hook_order=None):
            sa.actions = actions.data_ptr()
            sa.hook_order = hook_order.data_ptr() if hook_order is not None else None
            if phase is not None:               # (the staging protocol counts the steps: one phase per step, whoever issues it)
                gen_c.stage.phase = phase[0]
                phase[0] = (phase[0] + 1) & 0x3fffffff
            if _cur_device() == index:
                rc = fn(spec_ref, B, args_ref, _stream(dev))
            else:
                with device_ctx(dev):
                    rc = fn(spec_ref, B, args_ref, _stream(dev))
            if rc:
                _lib.check(rc, "mgx_step_ex")
            if side and phase[0] % every == 0:
                if gen_stream is not None:          # beside the steps: its own stream, ordered behind this step
                    gen_stream.wait_stream(torch.cuda.current_stream(dev))
                    handle = gen_stream.cuda_stream
                else:                               # between the steps, every `every`-th one: the steps' own stream
                    handle = _stream(dev)
                with device_ctx(dev):
                    rc = stage_fn(spec_ref, B, gen_ref, rng_ptr, episode_ptr, handle)
                if rc:
                    _lib.check(rc, "mgx_stage_generate")
                stage["launches"] = stage.get("launches", 0) + 1
        step._keep = (sa, keep)
        return step

    def stage_generate(self, B, gen, rng, episode):
        """mgx_stage_generate on torch's current stream: the staging slots' generator as a launch of its own (include/mgx.h)."""
        gen_c = self._layout_gen_struct(gen)
        with torch.cuda.device(rng.device):
            rc = _lib.lib().mgx_stage_generate(C.byref(self.sc), B, C.byref(gen_c), rng.data_ptr(), episode.data_ptr(), _stream(rng.device))
        _lib.check(rc, "mgx_stage_generate")
        gen["stage"]["launches"] = gen["stage"].get("launches", 0) + 1

    def bind_chains(self, B, parts, streams, grid, agents, rng, step_count, target, err, obs, dirs, reward, terminated,
                    truncated, auto_reset=None, one_hot: bool = False, generate=None):
        """The same step issued as `parts` launches over consecutive blocks of the batch on `streams` (mgx_step_chains): the
        chains of consecutive calls run independently.  Returns f(actions, fork_event, hook_order=None); nothing is joined."""
        fn = _lib.lib().mgx_step_chains
        sa, keep = self.step_args(grid, agents, rng, step_count, target, err, obs, dirs, reward, terminated, truncated,
                                  auto_reset, one_hot, generate)
        spec_ref, args_ref = C.byref(self.sc), C.byref(sa)
        handles = (C.c_void_p * parts)(*[s.cuda_stream for s in streams])
        dev = grid.device

        gen_c = keep[1]
        phase = generate[0]["stage"]["phase"] if generate is not None and generate[0].get("stage") is not None else None
        cand = phase is not None and bool(gen_c.stage.candidates)
        if cand:
            # candidates are made by generator launches BETWEEN the steps: every lead/2 steps the chains are joined on the first
            # stream, ONE generator launch serves the whole batch there, and the chains go on behind it
            stage = generate[0]["stage"]
            every = max(1, int(stage.get("lead", 2)) // 2)
            stage_fn, gen_ref = _lib.lib().mgx_stage_generate, C.byref(gen_c)
            rng_ptr, episode_ptr = rng.data_ptr(), generate[1].data_ptr()
        if phase is not None and not cand and gen_c.stage.external:
            # the chains issue no generator launches of their own (nobody sits between the steps of a chain): the pending snapshots are
            # served by generator wavefronts inside every chain's launch instead (MgxGenStage.external = 0, the in-launch form), so a
            # staged slot is there to adopt when the env truncates -- with `external` left set the requests would never be served
            gen_c.stage.external = 0

        def step(actions, fork_event, hook_order=None):
            sa.actions = actions.data_ptr()
            sa.hook_order = hook_order.data_ptr() if hook_order is not None else None
            if phase is not None:
                gen_c.stage.phase = phase[0]
                phase[0] = (phase[0] + 1) & 0x3fffffff
            with torch.cuda.device(dev):
                rc = fn(spec_ref, B, args_ref, parts, handles, fork_event)
                if rc:
                    _lib.check(rc, "mgx_step_chains")
                if cand and phase[0] % every == 0:
                    for s in streams[1:]:
                        streams[0].wait_stream(s)
                    rc = stage_fn(spec_ref, B, gen_ref, rng_ptr, episode_ptr, handles[0])
                    if rc:
                        _lib.check(rc, "mgx_stage_generate")
                    for s in streams[1:]:
                        s.wait_stream(streams[0])
                    stage["launches"] = stage.get("launches", 0) + 1
        step._keep = (sa, keep, handles, streams)
        return step

    def sub_shards(self, B, auto_reset: bool = False, one_hot: bool = False, generate: bool = False) -> int:
        """mgx_sub_shards: how many independent chains the step of `B` envs is best issued as on this device."""
        sa = _lib.MgxStepArgs()
        sa.steps, sa.one_hot = 1, int(bool(one_hot))
        ar = _lib.MgxAutoReset()
        if auto_reset and not generate:
            sa.auto_reset = C.pointer(ar)            # (only its presence matters: it selects the kernel instantiation)
        out = C.c_int32(1)
        with torch.cuda.device(self.device):
            rc = _lib.lib().mgx_sub_shards(C.byref(self.sc), B, C.byref(sa), C.byref(out))
        _lib.check(rc, "mgx_sub_shards")
        return int(out.value)

    def rollout(self, B, T, grid, agents, rng, step_count, actions, target, err, obs, dirs, reward, terminated,
                truncated, auto_reset=None, one_hot: bool = False, generate=None):
        """T steps in one launch (mgx_step_ex with steps = T); `one_hot`: obs is u8[T,B,A,v,v,21]; `generate` = (gen dict,
        episode, was_reset[T,B]): episodes that end are regenerated on the device (then T launches of the step kernel)."""
        sa, keep = self.step_args(grid, agents, rng, step_count, target, err, obs, dirs, reward, terminated, truncated,
                                  auto_reset, one_hot, generate)
        sa.steps = T
        sa.actions = actions.data_ptr()
        with torch.cuda.device(grid.device):
            rc = _lib.lib().mgx_step_ex(C.byref(self.sc), B, C.byref(sa), _stream(grid.device))
        _lib.check(rc, "mgx_rollout")
        if generate is not None and generate[0].get("stage") is not None:      # (T launches: one phase each, and the generator
                                                                                # launches between them: mgx_kernels.hip)
            ph = generate[0]["stage"]["phase"]
            ph[0] = (ph[0] + T) & 0x3fffffff

    # ------------------------------------------------------------------------------------------ persistent stepping
    def persistent_waves(self, B, auto_reset: bool = False) -> int:
        """mgx_persistent_waves: wavefronts of the persistent launch for B envs (raises MgxError(UNSUPPORTED) when they cannot
        all be resident at once)."""
        sa = _lib.MgxStepArgs()
        sa.steps = 1
        ar = _lib.MgxAutoReset()
        if auto_reset:
            sa.auto_reset = C.pointer(ar)
        out = C.c_int32(0)
        with torch.cuda.device(self.device):
            rc = _lib.lib().mgx_persistent_waves(C.byref(self.sc), B, C.byref(sa), C.byref(out))
        _lib.check(rc, "mgx_persistent_waves")
        return int(out.value)

    def persistent_struct(self, granules, done, ctrl, max_steps: int, timeout_ms: int):
        return _lib.MgxPersistent(granules.data_ptr(), done.data_ptr(), ctrl.data_ptr(), int(max_steps), int(timeout_ms))

    def persistent_launch(self, B, sa, pers, stream_handle: int):
        with torch.cuda.device(self.device):
            rc = _lib.lib().mgx_step_persistent(C.byref(self.sc), B, C.byref(sa), C.byref(pers), stream_handle)
        _lib.check(rc, "mgx_step_persistent")

    def persistent_post(self, B, actions, step: int, granules):
        dev = granules.device
        if _cur_device() == dev.index:
            rc = _lib.lib().mgx_persistent_post(C.byref(self.sc), B, actions.data_ptr(), step, granules.data_ptr(), _stream(dev))
        else:
            with torch.cuda.device(dev):
                rc = _lib.lib().mgx_persistent_post(C.byref(self.sc), B, actions.data_ptr(), step, granules.data_ptr(), _stream(dev))
        if rc:
            _lib.check(rc, "mgx_persistent_post")

    def persistent_wait(self, done, waves: int, step: int, ctrl, timeout_ms: int):
        dev = done.device
        if _cur_device() == dev.index:
            rc = _lib.lib().mgx_persistent_wait(done.data_ptr(), waves, step, ctrl.data_ptr(), timeout_ms, _stream(dev))
        else:
            with torch.cuda.device(dev):
                rc = _lib.lib().mgx_persistent_wait(done.data_ptr(), waves, step, ctrl.data_ptr(), timeout_ms, _stream(dev))
        if rc:
            _lib.check(rc, "mgx_persistent_wait")

    def persistent_feed(self, B, actions, steps: int, pers, waves: int, trace, stream_handle: int):
        with torch.cuda.device(self.device):
            rc = _lib.lib().mgx_persistent_feed(C.byref(self.sc), B, actions.data_ptr(), steps, C.byref(pers), waves,
                                                trace.data_ptr() if trace is not None else None, stream_handle)
        _lib.check(rc, "mgx_persistent_feed")

    def one_hot(self, cells, out):
        _one_hot_into(cells, out)

    def full_obs(self, B, grid, agents, out):
        _full_obs_into(self.sc, B, grid, agents, out)

    def reset_done(self, B, first_env, pool, grid, agents, step_count, target, episode, was_reset):
        pg, pa, pt = pool
        with torch.cuda.device(grid.device):
            rc = _lib.lib().mgx_reset_done(
                C.byref(self.sc), B, first_env, pg.shape[0], pg.data_ptr(), pa.data_ptr(),
                pt.data_ptr() if pt is not None else None, grid.data_ptr(), agents.data_ptr(), step_count.data_ptr(),
                target.data_ptr() if pt is not None else None, episode.data_ptr(), was_reset.data_ptr(),
                _stream(grid.device))
        _lib.check(rc, "mgx_reset_done")

    def reset_generate(self, B, gen, grid, agents, rng, step_count, aux, episode, was_reset):
        """gen = dict(kind, room_size, start=(x, y, dir), blank i16[H,W] (packed cells), gen_state i64[B,6]) -- include/mgx.h MgxLayoutGen"""
        g = self._layout_gen_struct(gen)
        with torch.cuda.device(grid.device):
            rc = _lib.lib().mgx_reset_generate(
                C.byref(self.sc), B, C.byref(g), grid.data_ptr(), agents.data_ptr(), rng.data_ptr(), step_count.data_ptr(),
                aux.data_ptr() if aux is not None else None, episode.data_ptr(),
                was_reset.data_ptr() if was_reset is not None else None, _stream(grid.device))
        _lib.check(rc, "mgx_reset_generate")

    def launch_info(self, B) -> dict:
        return _lib.launch_info(self.spec, B)
