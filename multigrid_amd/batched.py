"""BatchedMultiGridEnv: B independent MultiGrid environments resident in HBM, stepped by one fused HIP kernel.

This is the tensor-level form of `multigrid.base.MultiGridEnv` (multigrid/base.py:36-841 of the reference):
`reset` / `step` / `gen_obs` keep the reference's meaning, with dict-of-agents replaced by a leading
(batch, agent) shape.  The per-env dict API lives in multigrid_amd/env.py on top of this class.

State (device tensors, layouts in include/mgx.h):
    cells i16[B,H,W] (packed grid cells, include/mgx.h MgxCell; `grid` = the same as (type, color, state) bytes
    u8[B,H,W,3])   agents u8[B,A,8]   rng i64[B,4] (PCG64 words)   step_count i32[B]   aux u8[B,16]
Outputs of `step` (pre-allocated, overwritten by every call -- clone what you keep):
    obs u8[B,A,v,v,3]  dir u8[B,A]  reward f64[B,A]  terminated u8[B,A]  truncated u8[B]
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import layouts, rng as rnglib
from .constants import NO_ACTION, Action
from .spec import EnvSpec

INT32_MAX = 2 ** 31 - 1


class StepGraph:
    """A captured hipGraph of env steps (BatchedMultiGridEnv.capture_steps)."""

    def __init__(self, env, graph, sub_shards, keep):
        self.env, self.graph, self.sub_shards = env, graph, sub_shards
        self._version = env._layout_version
        self._keep = keep                # the sub-shards' pre-bound launchers, their streams, the action tensors

    def replay(self):
        if self.env._layout_version != self._version:
            raise RuntimeError("this graph was captured before set_layout_pool() / set_layout_generator() replaced the layout "
                               "tensors it holds pointers to: capture_steps() again")
        self.graph.replay()


class PersistentSession:
    """Closed-loop stepping without a kernel boundary per step (include/mgx.h: mgx_step_persistent): ONE launch of the step kernel
    stays resident on a stream of its own, keeps the env state in LDS, takes each step's actions as tagged granules and publishes
    a per-wavefront flag behind the step's outputs.  For batches in the latency regime (C2, C4's 8-GPU share, C3), where the launch
    boundary and the state reload are a third of a step.  Made by `BatchedMultiGridEnv.persistent()`; use as a context manager.

        obs, dirs = env.gen_obs()                 # (the first observation: BEFORE the session takes the state over)
        with env.persistent(max_steps=T) as ps:
            for t in range(T):
                obs, dirs, reward, terminated, truncated = ps.step(policy(obs))      # == env.step(...) bit for bit

    `ps.step(a)` = `ps.post(a)` (the actions, stream-ordered behind the policy that produced them) + `ps.wait()` (work enqueued on
    the current stream afterwards sees the step's outputs in the env's output buffers).  The env's own `step()` / `rollout()` /
    state tensors are unavailable until the session is closed (the state lives in the launch).  Inside a session synchronise
    STREAMS (`torch.cuda.current_stream().synchronize()`, events), never the device: `torch.cuda.synchronize()` waits for the
    resident launch itself, which waits for the next actions -- until its timeout ends the session."""

    def __init__(self, env, max_steps: int, auto_reset: bool, timeout_ms: int):
        be = env.backend
        if not hasattr(be, "persistent_launch"):
            raise RuntimeError("persistent stepping needs the HIP backend")
        if auto_reset and getattr(env, "_gen", None) is not None:
            raise RuntimeError("persistent stepping restarts finished envs from a layout pool (set_layout_pool), not from the "
                               "device generator")
        self.env, self.max_steps, self.timeout_ms = env, int(max_steps), int(timeout_ms)
        self.auto_reset = bool(auto_reset)
        B, A, dev = env.batch, env.spec.num_agents, env.device
        self.waves = be.persistent_waves(B, self.auto_reset)
        self.granules = torch.zeros((B, (A + 3) // 4), dtype=torch.int64, device=dev)
        self.done = torch.zeros((self.waves,), dtype=torch.int32, device=dev)
        self.ctrl = torch.tensor([0, 0, 0, -1] + [0] * 4, dtype=torch.int32, device=dev)
        # a stream of its own, HIGH priority: the launch runs BESIDE whatever produces its actions (HIP multiplexes the streams of
        # one priority over a few hardware queues; the priorities have queues of their own, so no producer stream ever shares
        # one with the launch), and its few wavefronts are served first wherever they share a SIMD with the producer's
        self.stream = torch.cuda.Stream(dev, priority=-1)
        self.t = 0                       # steps posted
        self.waited = 0                  # steps waited for
        self.open = False
        ar = env._auto_reset_args(self.auto_reset, getattr(env, "was_reset", None))
        self._sa, self._keep = be.step_args(env.cells, env.agents, env.rng, env.step_count,
                                            env.aux if env.spec.env_kind != "empty" else None, env.err, env.obs, env.dir,
                                            env.reward, env.terminated, env.truncated, auto_reset=ar)
        self._pers = be.persistent_struct(self.granules, self.done, self.ctrl, self.max_steps, self.timeout_ms)

    def __enter__(self):
        env = self.env
        env._need_state()
        env.join()
        if env._session is not None:
            raise RuntimeError("this env already has a persistent session open")
        cur = torch.cuda.current_stream(env.device)
        self.stream.wait_stream(cur)                       # (the state and the zeroed hand-shake words are in place)
        env.backend.persistent_launch(env.batch, self._sa, self._pers, self.stream.cuda_stream)
        env._session = self
        self.open = True
        return self

    def post(self, actions: torch.Tensor):
        """Hand over the actions of the next step (i8[B,A] on the env's device), stream-ordered on the current stream."""
        env = self.env
        if not self.open or self.t >= self.max_steps:
            raise RuntimeError("the persistent session is closed or has used up its max_steps")
        if actions.dtype is not torch.int8 or actions.shape != env._act_shape or actions.device != env.cells.device \
                or not actions.is_contiguous():
            raise ValueError(f"actions must be a contiguous int8 tensor of shape {tuple(env._act_shape)} on {env.cells.device}")
        self.t += 1
        env.backend.persistent_post(env.batch, actions, self.t, self.granules)

    def wait(self):
        """Work enqueued on the current stream after this call sees the outputs of the last posted step."""
        self.env.backend.persistent_wait(self.done, self.waves, self.t, self.ctrl, self.timeout_ms)
        self.waited = self.t
        env = self.env
        return env.obs, env.dir, env.reward, env.terminated, env.truncated

    def step(self, actions: torch.Tensor):
        self.post(actions)
        return self.wait()

    def feed(self, actions: torch.Tensor, trace: bool = False):
        """A recorded action sequence i8[T,B,A] played through the closed-loop hand-shake by ONE resident workgroup
        (mgx_persistent_feed: the shortest producer there can be; benchmarks and tests).  Returns the trace tensor
        (i64[2T + 1], s_memrealtime ticks of 10 ns) or None.  The session's remaining steps must cover T."""
        env = self.env
        T = int(actions.shape[0])
        if self.t != 0 or T != self.max_steps:
            raise RuntimeError("feed() plays a whole session: max_steps == len(actions), nothing posted before")
        if actions.dtype is not torch.int8 or tuple(actions.shape[1:]) != tuple(env._act_shape) or not actions.is_contiguous():
            raise ValueError("actions must be a contiguous int8 tensor [T,B,A]")
        tr = torch.zeros((2 * T + 1,), dtype=torch.int64, device=env.device) if trace else None
        env.backend.persistent_feed(env.batch, actions, T, self._pers, self.waves, tr, torch.cuda.current_stream(env.device).cuda_stream)
        self.t = self.waited = T
        return tr

    def close(self):
        """End the launch (a stop request unless it has run all of its steps), join its stream: the env's state tensors hold
        the state after the last completed step.  Raises if a wait inside the launch timed out."""
        if not self.open:
            return
        env = self.env
        cur = torch.cuda.current_stream(env.device)
        if self.t < self.max_steps:
            if self.waited < self.t:
                self.wait()
            self.ctrl[0:1].fill_(1)                         # stop request, stream-ordered behind the last wait
        cur.wait_stream(self.stream)
        self.open = False
        env._session = None
        c = [int(v) & 0xFFFFFFFF for v in self.ctrl.cpu()]
        self.timeouts, self.waves_left, self.steps_completed = c[1], c[2], c[3]
        if c[1]:
            raise RuntimeError(f"persistent stepping: {c[1]} wait(s) timed out after {self.timeout_ms} ms "
                               f"(steps completed by every wavefront: {c[3]})")

    def __exit__(self, *exc):
        try:
            self.close()
        except RuntimeError:
            if exc[0] is None:
                raise
        return False

    def __del__(self):
        # a session dropped without close(): ask the launch to stop (it would otherwise wait out its timeout on the device); the env
        # stays marked as having a session -- its state tensors are only valid once somebody has joined the launch's stream
        try:
            if getattr(self, "open", False):
                self.ctrl[0:1].fill_(1)
        except Exception:               # noqa: BLE001  (interpreter shutdown, device gone: nothing to do)
            pass


class BatchedMultiGridEnv:
    def __init__(self, spec: EnvSpec, batch: int, device="cuda", *, first_env: int = 0, backend=None, specialise=None):
        """
        spec       environment class configuration
        batch      number of envs held by THIS process (its shard of the global batch)
        device     HIP device ('cuda', 'cuda:3', ...).  CPU devices are refused: there is no CPU path.
        first_env  global index of env 0 of this shard; per-env seeds are a function of the global index so
                   results do not depend on how the batch is sharded over GPUs
        backend    launcher override used by the test-suite to exercise this host logic without a GPU;
                   product code leaves it None (= ops.HipBackend).
        specialise True: compile a shape-specialised step kernel for this (spec, batch) at run time if the library has none built
                   in (multigrid_amd/jit.py: hipRTC, ~1-2 s once, cached on disk; 10-14 % of a step in the latency regime, same
                   results).  None (default): what the environment variable MGX_JIT says (unset / 0 = no).
        """
        self.spec = spec
        self.batch = int(batch)
        self.first_env = int(first_env)
        self.device = torch.device(device)
        if backend is None:
            from .ops import HipBackend
            backend = HipBackend(spec, self.device)
        self.backend = backend
        B, A, dev = self.batch, spec.num_agents, self.device
        # packed cells: MgxCell bit patterns (i16), or -- spec.cell_bytes == 1 -- the compact MgxCell8 bytes (include/mgx.h)
        # (spec.cell_bytes == 3: the reference's byte triples themselves, u8[B,H,W,3])
        self.cells = torch.zeros(spec.cells_shape(B), dtype=torch.uint8 if spec.cell_bytes != 2 else torch.int16, device=dev)
        self.agents = torch.zeros(spec.agents_shape(B), dtype=torch.uint8, device=dev)
        self.rng = torch.zeros((B, 4), dtype=torch.int64, device=dev)
        self.step_count = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.aux = torch.zeros((B, 16), dtype=torch.uint8, device=dev)       # the env subclass' hook state (include/mgx.h)
        # Everything a step hands back -- reward, obs, dir, terminated, truncated, the error words -- lies in ONE allocation (16-byte
        # aligned views of it), so that a host caller that wants all of it pays one device-to-host copy, not six (`outputs_to_host`:
        # the dict API, multigrid_amd/env.py; round 6)
        obs_shape = tuple(spec.obs_shape(B))
        parts = (("reward", torch.float64, (B, A)), ("obs", torch.uint8, obs_shape), ("dir", torch.uint8, (B, A)),
                 ("terminated", torch.uint8, (B, A)), ("truncated", torch.uint8, (B,)), ("err", torch.int32, (2,)))
        offs, total = {}, 0
        for name, dt, shape in parts:
            offs[name] = total
            total += (int(np.prod(shape)) * torch.empty((), dtype=dt).element_size() + 15) & ~15
        self._out = torch.zeros(max(total, 16), dtype=torch.uint8, device=dev)
        self._out_parts, self._out_host = [(n, dt, sh, offs[n]) for n, dt, sh in parts], None
        for name, dt, shape in parts:
            nbytes = int(np.prod(shape)) * torch.empty((), dtype=dt).element_size()
            setattr(self, name, self._out[offs[name]:offs[name] + nbytes].view(dt).view(shape))
        self.err.copy_(torch.tensor([0, INT32_MAX], dtype=torch.int32))
        self._loaded = False
        self._act_shape = torch.Size((B, A))
        self._bound = {}                 # (auto_reset, one_hot, generate[, parts]) -> pre-bound step launcher (ops.HipBackend)
        self._layout_version = 0         # bumped whenever the layout pool / generator tensors are REPLACED: sub-shards and
        self._parent = None              # captured graphs made before hold pointers into the old ones and refuse to run
        self._chain_streams = []         # side streams of the eager sub-shard form (step(..., sub_shards=P))
        self._chains_pending, self._chains_P = False, 1
        self._session = None             # an open PersistentSession: the state lives in its launch
        self.shape_kernel = None
        if specialise is None:
            specialise = os.environ.get("MGX_JIT", "0") not in ("", "0")
        if specialise and self.device.type == "cuda" and getattr(self.backend, "name", "") == "hip":
            self.specialise()

    def specialise(self) -> str:
        """Run this env's step on a kernel compiled for exactly its shape (multigrid_amd.jit.ensure_shape).  Returns what happened:
        "built-in" / "registered" / "compiled" / "not-latency" / "unavailable" (also kept in `.shape_kernel`)."""
        from . import jit
        self.shape_kernel = jit.ensure_shape(self.spec, self.batch, self.device)
        return self.shape_kernel

    @property
    def grid(self) -> torch.Tensor:
        """The grid as (type, color, state) bytes u8[B,H,W,3] ([y][x]: layouts.grid_from_product gives the reference's
        Grid.state) -- unpacked from `cells` on every access; for inspection, tests and checkpoints, not the hot path.  A box that
        holds something carries it in the upper bits of its state byte (include/mgx.h "BOX CONTENTS"; `& 3` is what Grid.state
        shows)."""
        if self.spec.cell_bytes == 3:
            return self.cells
        c = self.cells.to(torch.int32)
        if self.spec.cell_bytes == 1:                        # MgxCell8: joint (type, state) code | color << 4 | opaque << 7
            tc = c & 0xF
            door = (tc == 11) | (tc == 12)
            t = torch.where(tc <= 10, tc, torch.where(door, torch.full_like(tc, 4), torch.full_like(tc, 10)))
            st = torch.where(tc <= 10, torch.zeros_like(tc), torch.where(door, tc - 10, tc - 12))
            return torch.stack((t, (c >> 4) & 0x7, st), dim=-1).to(torch.uint8)
        content = ((c >> 4) & 0xF) | (((c >> 11) & 1) << 4) | (((c >> 14) & 1) << 5)      # a box's content (include/mgx.h)
        return torch.stack((c & 0xF, (c >> 8) & 0x7, ((c >> 12) & 0x3) | (content << 2)), dim=-1).to(torch.uint8)

    # ------------------------------------------------------------------------------------------ state in
    def load_state(self, grid, agents, rng=None, aux=None, step_count=None, validate: bool = True, target=None):
        """Install an initial state (numpy arrays or tensors in the product layout).  A single env's state
        (no batch dim) is broadcast to the whole batch.  This is also the parity-injection point: the
        reference's `Grid.state` / `AgentState` go through layouts.grid_to_product / pack_agents."""
        self._no_session("load_state")
        sp, B = self.spec, self.batch

        def prep(x, shape, dtype):
            t = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
            if t.dim() == len(shape) - 1:
                t = t.unsqueeze(0).expand(shape)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
            return t.to(dtype=dtype).contiguous()

        self.join()
        g = prep(grid, sp.grid_shape(B), torch.uint8)
        a = prep(agents, sp.agents_shape(B), torch.uint8)
        if validate:
            layouts.check_walled(g.cpu().numpy() if g.device.type != "cpu" else g.numpy())
            an = a.cpu().numpy()
            if (an[..., 2] >= sp.width).any() or (an[..., 3] >= sp.height).any() or (an[..., 1] > 3).any():
                raise ValueError("agent position / direction out of range")
        self._refuse_carried_contents(a)
        self.cells.copy_(torch.from_numpy(layouts.pack_cells_for(sp, g.cpu().numpy())))
        self.agents.copy_(a)
        if rng is not None:
            r = np.asarray(rng.cpu() if torch.is_tensor(rng) else rng)
            if r.dtype == np.uint64:
                r = r.view(np.int64)
            self.rng.copy_(prep(r, (B, 4), torch.int64))
        if aux is None and target is not None:                 # BlockedUnlockPickup convenience: (type, color, state[, 0])
            t = np.zeros(np.asarray(target).shape[:-1] + (16,), dtype=np.uint8)
            t[..., :min(4, np.asarray(target).shape[-1])] = np.asarray(target)[..., :4]
            aux = t
        if aux is not None:
            self.aux.copy_(prep(aux, (B, 16), torch.uint8))
        elif self.spec.env_kind != "empty":
            raise ValueError(f"env_kind {self.spec.env_kind!r} needs `aux` (the env's hook state, include/mgx.h)")
        if step_count is None:
            self.step_count.zero_()
        else:
            self.step_count.copy_(prep(step_count, (B,), torch.int32))
        self._reset_err()
        self._state_written()
        self._loaded = True

    def _refuse_carried_contents(self, agent_rows):
        """Compact cells (EnvSpec.cell_bytes = 1) have no room for what a box holds (include/mgx.h: MgxCell8): a filled box on the
        GRID is refused by the packer; one in an agent's HANDS (carry state byte >> 2, include/mgx.h "BOX CONTENTS") would lose its
        content when it is put down -- refused here, so that both formats give the same results or an error."""
        if self.spec.cell_bytes == 1:
            rows = agent_rows.cpu().numpy() if torch.is_tensor(agent_rows) else np.asarray(agent_rows)
            if (rows[..., 7] >> 2).any():
                raise ValueError("compact cells (cell_bytes = 1) cannot hold a box's content: an agent carries a filled box; "
                                 "use cell_bytes = 2 for envs with Box(contains=...)")

    def seed(self, seed: int):
        """Per-env `np_random = Generator(PCG64(SeedSequence([seed, global_env_index])))`; global env 0 gets
        `SeedSequence(seed)`, exactly what `gym.Env.reset(seed=seed)` gives the reference's single env
        (multigrid/base.py:269).  Keyed on the pair so that consecutive experiment seeds share no stream (seed + index
        would give env b under seed s+1 the stream of env b+1 under seed s), and a pure function of the GLOBAL env index
        so that sharding does not change any env's stream."""
        self._no_session("seed")
        self.join()
        idx = self.first_env + np.arange(self.batch)
        self.rng.copy_(torch.from_numpy(rnglib.words_from_seed_and_index(seed, idx).view(np.int64)))
        self._state_written()

    def seed_synthetic(self, seed: int):
        """Benchmark-grade seeding: valid PCG64 states from a hash of the global env index (fast for large B)."""
        self._no_session("seed_synthetic")
        self.join()
        words = rnglib.synthetic_words(self.batch, seed, self.first_env)
        self.rng.copy_(torch.from_numpy(words.view(np.int64)))
        self._state_written()

    def _reset_err(self):
        self.err.copy_(torch.tensor([0, INT32_MAX], dtype=torch.int32))

    def _state_written(self):
        """The env state (np_random, step counts, grids) was replaced from outside: what the staged generator prepared from the
        old state is void (the slots are a cache: dropping them is always valid), and nothing of the old state is in flight."""
        st = (getattr(self, "_gen", None) or {}).get("stage")
        if st is not None and st.get("candidates"):
            st["tag"].fill_(-1)
        elif st is not None:
            st["tag"][:, 0] = -1
            st["tag"][:, 3] = 0

    # ------------------------------------------------------------------------------------------ hot path
    def gen_obs(self, one_hot: bool = False):
        """multigrid/base.py:348-376 for every env: returns (obs u8[B,A,v,v,3], dir u8[B,A]).
        one_hot=True: the observation comes out one-hot encoded, u8[B,A,v,v,21] (`OneHotObsWrapper`,
        multigrid/wrappers.py:158-190), written by the same kernel launch; `obs` is not touched."""
        self._need_state()
        self.join()
        if one_hot:
            if self.spec.cell_bytes == 1:          # compact cells: the one-hot STEP is one launch (round 6); the observation of a
                self.backend.gen_obs(self.batch, self.cells, self.agents, self.obs, self.dir)      # reset is two: gen_obs, then
                self.backend.one_hot(self.obs, self._one_hot_buffer())                             # the one-hot kernel over it
                return self._one_hot, self.dir
            self._need_wide_cells("one-hot output")
            self.backend.gen_obs(self.batch, self.cells, self.agents, self._one_hot_buffer(), self.dir, one_hot=True)
            return self._one_hot, self.dir
        self.backend.gen_obs(self.batch, self.cells, self.agents, self.obs, self.dir)
        return self.obs, self.dir

    def _one_hot_buffer(self):
        if getattr(self, "_one_hot", None) is None:
            self._one_hot = torch.zeros(tuple(self.obs.shape[:-1]) + (21,), dtype=torch.uint8, device=self.device)
        return self._one_hot

    def _auto_reset_args(self, auto_reset, was_reset):
        if not auto_reset:
            return None
        if getattr(self, "_pool", None) is None:
            raise RuntimeError("auto_reset needs set_layout_pool() first")
        return (self.first_env, self._pool, self.episode, was_reset)

    def step(self, actions: torch.Tensor, auto_reset: bool = False, one_hot: bool = False, hook_order=None, sub_shards=1):
        """multigrid/base.py:303-346 for every env.

        one_hot=True: the first element returned is the one-hot observation u8[B,A,v,v,21] (what RLlib's default
        `OneHotObsWrapper` registration feeds the policy, multigrid/rllib/__init__.py:110-111), produced by the same launch
        (mgx_step_one_hot); the 3-channel `obs` buffer is not written.

        auto_reset=True fuses `reset_done()` into the launch (build-defined, include/mgx.h mgx_step_autoreset): an env
        whose episode ended with the previous step first restarts from the layout pool, then takes this step's
        actions; `was_reset` tells which ones did.  Bit-identical to `reset_done(); step(actions)`.

        actions     i8[B,A] on the env's device; `Action` values 0..6, NO_ACTION (-1) = agent not acting
                    (its key absent from the reference's actions dict, base.py:403-404).
        hook_order  u8[B,A] or None: per env, agent indices in the insertion order of the caller's actions dict -- the order
                    in which the RedBlueDoors / LockedHallway step hooks visit the agents (`for agent_id, action in
                    actions.items()`, multigrid/envs/redbluedoors.py:176, locked_hallway.py:210).  None = ascending index.
        sub_shards  1 (default): one launch for the whole batch on the current stream.  "auto" = `sub_shards_hint(form="eager")`,
                    which is 1 (per-step calls from Python are host-bound as chains); P > 1: the step is issued as P launches over consecutive blocks
                    of the batch on P side streams (mgx_step_chains) and is NOT joined -- consecutive calls form P independent
                    chains, so one block's load phase runs under another's compute (C4: 20.9 -> ~16 us per step).  The
                    outputs / state are complete after `join()`; any other method of this object joins first.  Same
                    results bit for bit (envs are independent; seeds and layouts follow the global env index).
        Returns (obs, dir, reward, terminated, truncated) -- the env's output buffers.
        An unknown action value does not raise here (no device sync on the hot path); it is recorded in
        `err` and surfaced as ValueError by `check_errors()` (multigrid/base.py:473-474).
        """
        if not self._loaded or self._session is not None:
            self._need_state()
        if self._parent is not None:
            self._check_fresh()
        if actions.dtype is not torch.int8 or actions.shape != self._act_shape or actions.device != self.cells.device \
                or not actions.is_contiguous():
            raise ValueError(f"actions must be a contiguous int8 tensor of shape {tuple(self._act_shape)} "
                             f"on {self.cells.device}")
        if sub_shards == 1 and hook_order is None and not self._chains_pending:      # the hot path: one dict lookup, one call
            fast = self._bound.get((auto_reset, one_hot))
            if fast is not None:
                fast[0](actions)
                return fast[1]
        if hook_order is not None and (hook_order.dtype is not torch.uint8 or hook_order.shape != self._act_shape
                                       or hook_order.device != self.cells.device or not hook_order.is_contiguous()):
            raise ValueError(f"hook_order must be a contiguous uint8 tensor of shape {tuple(self._act_shape)} on {self.cells.device}")
        if one_hot and not (self.spec.cell_bytes == 1 and self.spec.env_kind == "empty"):
            self._need_wide_cells("one-hot output")        # (compact cells: the hook-free step has a one-hot instantiation, round 6)
        generate = bool(auto_reset) and getattr(self, "_gen", None) is not None
        if generate:
            # on-device generation: the envs whose episode ends with THIS step are regenerated right after it (in the tail
            # of the same launch when the launcher can: mgx_step_generate); `was_reset` = the envs regenerated by this call
            auto_reset = False
        P = 1
        if sub_shards != 1:
            P = self.sub_shards_hint(auto_reset or generate, one_hot, form="eager") if sub_shards == "auto" else int(sub_shards)
            P = max(1, min(P, self.batch // self.SUB_SHARD_ALIGN))
        if self._chains_pending and P != self._chains_P:       # (another cut of the batch over other streams: join the old chains)
            self.join()
        key = (bool(auto_reset), bool(one_hot), generate, P)
        if one_hot:
            self._one_hot_buffer()
        fast = self._bound.get(key)
        if fast is None:
            fast = self._bind_step(*key)
        out = (self._one_hot if one_hot else self.obs), self.dir, self.reward, self.terminated, self.truncated
        if fast is not False:
            if P > 1:
                ev = self._fork_event
                ev.record(torch.cuda.current_stream(self.device))
                fast(actions, ev.cuda_event, hook_order)
                # the chains read `actions` / `hook_order` on their own streams and are not joined: the caching allocator must
                # not hand a temporary's block to the next allocation of the current stream while they still read it
                for st in self._chain_streams[:P]:
                    actions.record_stream(st)
                    if hook_order is not None:
                        hook_order.record_stream(st)
                self._chains_pending, self._chains_P = True, P
                return out
            if isinstance(auto_reset, bool) and isinstance(one_hot, bool):
                self._bound[(auto_reset or generate, one_hot)] = (fast, out)       # (the hot path's entry: cleared with _bound)
            fast(actions, hook_order)
            return out
        # launchers without bind_step (the test-suite's oracle backend)
        sp = self.spec
        ar = self._auto_reset_args(auto_reset, getattr(self, "was_reset", None))
        kw = {}
        if ar is not None:
            kw["auto_reset"] = ar
        if one_hot:
            kw["one_hot"] = True
        if hook_order is not None:
            kw["hook_order"] = hook_order
        obs = self._one_hot_buffer() if one_hot else self.obs
        self.backend.step(self.batch, self.cells, self.agents, self.rng, self.step_count, actions,
                          self.aux if sp.env_kind != "empty" else None, self.err,
                          obs, self.dir, self.reward, self.terminated, self.truncated, **kw)
        if generate:
            self.reset_done()
        return obs, self.dir, self.reward, self.terminated, self.truncated

    def join(self):
        """Make torch's current stream wait for the sub-shard chains started by `step(..., sub_shards=P)`: afterwards work
        enqueued on the current stream sees every block's outputs and state.  (No host synchronisation.)"""
        if self._chains_pending:
            cur = torch.cuda.current_stream(self.device)
            for st in self._chain_streams:
                cur.wait_stream(st)
            self._chains_pending = False

    def sub_shards_hint(self, auto_reset: bool = False, one_hot: bool = False, form: str = "graph") -> int:
        """How many independent chains this env's step is best issued as on its device.

        form="graph" (`capture_steps(sub_shards="auto")`): mgx_sub_shards -- 1 when a launch is less than two wavefronts per
        SIMD, else 2, from the kernel's occupancy and the device's CU count, not from constants (C4: 18.6 -> 16.0-16.7 us per step).
        form="eager" (`step(sub_shards="auto")`): 1.  Issued per step from Python, P launches + P event waits + the allocator's
        stream bookkeeping cost the host more than a step lasts (C4, two chains: 32 us per step, host-bound, against 18.7 us as one
        launch: profiles/r5_bench.json `eager.c4_chains`, rounds 3-5 alike), so the chains only pay where the host is out of the
        loop -- in a captured graph.  An explicit `step(sub_shards=P)` is still honoured."""
        if form not in ("graph", "eager"):
            raise ValueError("form must be 'graph' or 'eager'")
        if form == "eager":
            return 1
        hint = getattr(self.backend, "sub_shards", None)
        if hint is None:
            return 1
        generate = bool(auto_reset) and getattr(self, "_gen", None) is not None
        return hint(self.batch, bool(auto_reset), bool(one_hot), generate)

    def _bind_step(self, auto_reset: bool, one_hot: bool, generate: bool = False, parts: int = 1):
        """Resolve every pointer of the step call once (ops.HipBackend.bind_step / bind_chains); False when the launcher
        cannot."""
        key = (auto_reset, one_hot, generate, parts)
        bind = getattr(self.backend, "bind_step" if parts == 1 else "bind_chains", None)
        if bind is None:
            self._bound[key] = False
            return False
        ar = self._auto_reset_args(auto_reset, getattr(self, "was_reset", None))
        head = (self.batch,)
        if parts > 1:
            while len(self._chain_streams) < parts:
                self._chain_streams.append(torch.cuda.Stream(self.device))
            if getattr(self, "_fork_event", None) is None:
                self._fork_event = torch.cuda.Event()
                self._fork_event.record(torch.cuda.current_stream(self.device))     # (creates the handle)
            head = (self.batch, parts, self._chain_streams[:parts])
        f = bind(*head, self.cells, self.agents, self.rng, self.step_count,
                 self.aux if self.spec.env_kind != "empty" else None, self.err,
                 self._one_hot_buffer() if one_hot else self.obs, self.dir, self.reward, self.terminated, self.truncated,
                 auto_reset=ar, one_hot=one_hot,
                 **({"generate": (self._gen, self.episode, self.was_reset)} if generate else {}))
        self._bound[key] = f
        return f

    def _check_fresh(self):
        """A sub-shard (or a captured graph) holds pointers into its parent's layout pool / generator tensors."""
        if self._parent is not None and self._parent._layout_version != self._made_version:
            raise RuntimeError("this sub-shard was split off before its parent's set_layout_pool() / set_layout_generator() "
                               "replaced the layout tensors: split() / capture_steps() again")

    # ------------------------------------------------------------------------------------------ sub-shards
    SUB_SHARD_ALIGN = 64         # envs: keeps every sub-shard's tensors 16-byte aligned and its wavefronts' tiles as in the whole

    def split(self, parts: int) -> list:
        """`parts` sub-shards of this env: views over consecutive blocks of its tensors (nothing is copied; `first_env` follows,
        so seeds and auto-reset layouts stay functions of the global env index).  Envs are independent, so stepping the
        sub-shards -- in any interleaving, on any streams -- is stepping this env.  Call it after the state, the layout pool /
        generator and any one-hot use are set up (the views are taken of the tensors as they are now)."""
        self._need_state()
        self.join()
        B, al = self.batch, self.SUB_SHARD_ALIGN
        parts = max(1, min(int(parts), max(1, B // al)))
        cuts = [0] + [min(B, ((B * i // parts + al - 1) // al) * al) for i in range(1, parts)] + [B]
        out = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi <= lo:
                continue
            c = object.__new__(BatchedMultiGridEnv)
            c.spec, c.batch, c.first_env, c.device, c.backend = self.spec, hi - lo, self.first_env + lo, self.device, self.backend
            for name in ("cells", "agents", "rng", "step_count", "aux", "obs", "dir", "reward", "terminated", "truncated"):
                setattr(c, name, getattr(self, name)[lo:hi])
            c.err = self.err                                                  # (shared: the kernels update it atomically)
            c._loaded, c._act_shape, c._bound = True, torch.Size((hi - lo, self.spec.num_agents)), {}
            c._parent, c._made_version, c._layout_version = self, self._layout_version, 0
            c._chain_streams, c._chains_pending, c._chains_P, c._session = [], False, 1, None
            c._one_hot = self._one_hot[lo:hi] if getattr(self, "_one_hot", None) is not None else None
            c._pool = getattr(self, "_pool", None)
            c._gen = None
            if getattr(self, "_gen", None) is not None:
                c._gen = dict(self._gen, gen_state=self._gen["gen_state"][lo:hi])
                if self._gen.get("stage") is not None:          # (its own slice of the staging slots, its own step count)
                    st = self._gen["stage"]
                    c._gen["stage"] = {k: (v[lo:hi] if torch.is_tensor(v) else v) for k, v in st.items()}
                    if st.get("external") and st.get("stream") is not None:
                        c._gen["stage"]["stream"] = torch.cuda.Stream(self.device)
                    c._gen["stage"]["phase"] = [st["phase"][0]]
            if getattr(self, "episode", None) is not None:
                c.episode, c.was_reset = self.episode[lo:hi], self.was_reset[lo:hi]
            c._range = (lo, hi)
            out.append(c)
        return out

    def capture_steps(self, actions: torch.Tensor, auto_reset: bool = False, one_hot: bool = False, sub_shards=1,
                      hook_order=None):
        """A hipGraph of `len(actions)` consecutive `step` launches reading `actions[t]` (i8[T,B,A], kept by reference: refill
        it between replays; `hook_order` u8[T,B,A] likewise).  `graph.replay()` then costs no Python per step; the outputs of
        the LAST step are in the env's buffers.  (A policy in the loop is captured the same way: see examples/closed_loop.py.)

        sub_shards=P > 1 or "auto" (`sub_shards_hint()`): the batch is stepped as P independent sub-shards (`split`), each a
        chain of T launches on its own stream -- P parallel branches of the one graph.  A launch that fills the chip in a
        single round of wavefronts first loads (no wave has data to work on), then computes, then drains, and the next launch of
        its chain cannot start before its last wave has gone; the chains drift apart and fill each other's bubbles (C4: 20.9 ->
        15.7 us per step of the whole batch, C5 85 -> 69 us; in-kernel timestamps of the overlap: profiles/r3_chain_overlap.txt).
        Same results: envs are independent.  This is the double-buffered actor loop of a closed-loop caller (policy on one block
        while the others step).  A policy that needs all B observations of step t before step t+1 uses sub_shards=1.

        Returns a StepGraph (`.replay()`, `.graph` = the torch.cuda.CUDAGraph, `.sub_shards`); it refuses to replay once
        set_layout_pool() / set_layout_generator() have replaced the tensors it captured pointers to."""
        self._need_state()
        self.join()
        if sub_shards == "auto":
            sub_shards = self.sub_shards_hint(auto_reset, one_hot)
        stream = torch.cuda.current_stream(self.device)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(self.device)
        side.wait_stream(stream)
        if one_hot:
            self._one_hot_buffer()
        shards = self.split(sub_shards) if sub_shards > 1 else [self]
        others = [torch.cuda.Stream(self.device) for _ in shards[1:]]
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for s in others:                                             # fork
                    s.wait_stream(side)
                for i, sh in enumerate(shards):
                    lo, hi = getattr(sh, "_range", (0, self.batch))
                    with torch.cuda.stream(side if i == 0 else others[i - 1]):
                        st = (getattr(sh, "_gen", None) or {}).get("stage") if auto_reset else None
                        before = st.get("launches", 0) if st else 0
                        for t in range(actions.shape[0]):
                            ho = None if hook_order is None else (hook_order[t] if sh is self else hook_order[t, lo:hi])
                            sh.step(actions[t] if sh is self else actions[t, lo:hi], auto_reset=auto_reset, one_hot=one_hot,
                                    hook_order=ho)
                        # candidates (set_layout_generator): a block shorter than the generator launches' cadence may have caught
                        # none of them -- and would then never make a candidate, however often it is replayed: it ends with one
                        # (every shard makes the candidates of its own slice, on its own chain: ADVICE r5)
                        if st and st.get("candidates") and st.get("launches", 0) == before:
                            sh.backend.stage_generate(sh.batch, sh._gen, sh.rng, sh.episode)
                for i, sh in enumerate(shards):                              # (generator streams of side-staged envs: join)
                    gs = ((getattr(sh, "_gen", None) or {}).get("stage") or {}).get("stream")
                    if gs is not None:
                        (side if i == 0 else others[i - 1]).wait_stream(gs)
                for s in others:                                             # join
                    side.wait_stream(s)
        stream.wait_stream(side)
        return StepGraph(self, graph, len(shards), (shards, others, actions, hook_order))

    def rollout(self, actions: torch.Tensor, out: dict | None = None, auto_reset: bool = False, one_hot: bool = False) -> dict:
        """`T` consecutive `step`s in one kernel launch (env state stays in LDS between steps); bit-identical to
        calling `step(actions[t])` for t = 0..T-1.  For open-loop action sequences (random / scripted policies).

        actions  i8[T,B,A].  Returns {'obs': u8[T,B,A,v,v,3], 'dir', 'reward', 'terminated': [T,B,A],
        'truncated': u8[T,B]} (pass `out` to reuse buffers).  The env's own `obs`... buffers are not touched.
        auto_reset=True: finished envs restart from the layout pool before each step (as `step(auto_reset=True)`);
        the result then also holds 'was_reset': u8[T,B].  With a layout generator set (`set_layout_generator`) the envs whose
        episode ends with a step are regenerated on the device right after it instead, and the call is T launches of the
        step kernel (as T calls of `step(auto_reset=True)`).  one_hot=True: 'obs' is the one-hot observation u8[T,B,A,v,v,21]
        (multigrid/wrappers.py:158-190), written by the same launch."""
        self._need_state()
        generate = bool(auto_reset) and getattr(self, "_gen", None) is not None
        if not (self.spec.cell_bytes == 1 and self.spec.env_kind == "empty" and not one_hot and not generate):
            self._need_wide_cells("rollout (the steps of one launch)")      # (compact cells: the hook-free rollout, round 6)
        self.join()
        sp, B = self.spec, self.batch
        if actions.dtype != torch.int8 or actions.dim() != 3 or tuple(actions.shape[1:]) != (B, sp.num_agents) \
                or actions.device != self.cells.device or not actions.is_contiguous():
            raise ValueError(f"actions must be a contiguous int8 tensor of shape (T, {B}, {sp.num_agents}) on {self.cells.device}")
        T, A, v, dev = actions.shape[0], sp.num_agents, sp.view_size, self.device
        if out is None:
            out = {"obs": torch.empty((T, B, A, v, v, 21 if one_hot else 3), dtype=torch.uint8, device=dev),
                   "dir": torch.empty((T, B, A), dtype=torch.uint8, device=dev),
                   "reward": torch.empty((T, B, A), dtype=torch.float64, device=dev),
                   "terminated": torch.empty((T, B, A), dtype=torch.uint8, device=dev),
                   "truncated": torch.empty((T, B), dtype=torch.uint8, device=dev)}
        if auto_reset and "was_reset" not in out:
            out["was_reset"] = torch.empty((T, B), dtype=torch.uint8, device=dev)
        generate = bool(auto_reset) and getattr(self, "_gen", None) is not None
        if generate:        # episode starts generated on the device: the envs whose episode ends with step t are regenerated
            self.backend.rollout(B, T, self.cells, self.agents, self.rng, self.step_count, actions,          # right after it
                                 self.aux if sp.env_kind != "empty" else None, self.err, out["obs"], out["dir"],
                                 out["reward"], out["terminated"], out["truncated"],
                                 generate=(self._gen, self.episode, out["was_reset"]), **({"one_hot": True} if one_hot else {}))
            return out
        ar = self._auto_reset_args(auto_reset, out.get("was_reset"))
        self.backend.rollout(B, T, self.cells, self.agents, self.rng, self.step_count, actions,
                             self.aux if sp.env_kind != "empty" else None, self.err, out["obs"], out["dir"],
                             out["reward"], out["terminated"], out["truncated"],
                             **({"auto_reset": ar} if ar is not None else {}), **({"one_hot": True} if one_hot else {}))
        return out

    # ------------------------------------------------------------------------------------------ either side of the path
    def one_hot_obs(self) -> torch.Tensor:
        """`OneHotObsWrapper` (multigrid/wrappers.py:101-190) applied to the current `obs`: u8[B,A,v,v,21]."""
        self.join()
        self.backend.one_hot(self.obs, self._one_hot_buffer())
        return self._one_hot

    def full_obs(self) -> torch.Tensor:
        """`FullyObsWrapper` (multigrid/wrappers.py:17-58): u8[B,W,H,3], the grid in the reference's [x][y]
        orientation with every agent (terminated or not) drawn at its position."""
        self._need_state()
        self.join()
        if getattr(self, "_full", None) is None:
            self._full = torch.zeros((self.batch, self.spec.width, self.spec.height, 3), dtype=torch.uint8,
                                     device=self.device)
        self.backend.full_obs(self.batch, self.cells, self.agents, self._full)
        return self._full

    def set_layout_pool(self, grids, agents, auxs=None):
        """Pool of K pre-generated episode starts for `reset_done()`: u8[K,H,W,3], u8[K,A,8], u8[K,16] | None."""
        self._no_session("set_layout_pool")
        sp = self.spec
        g = torch.as_tensor(np.asarray(grids), dtype=torch.uint8)
        a = torch.as_tensor(np.asarray(agents), dtype=torch.uint8)
        K = g.shape[0]
        if tuple(g.shape) != (K, sp.height, sp.width, 3) or tuple(a.shape) != (K, sp.num_agents, 8) or K < 1:
            raise ValueError("layout pool has the wrong shape")
        layouts.check_walled(g.numpy())
        self._refuse_carried_contents(a)
        t = None
        if auxs is not None:
            t = torch.as_tensor(np.asarray(auxs), dtype=torch.uint8)
            if tuple(t.shape) != (K, 16):
                raise ValueError("layout pool aux must be u8[K,16]")
            t = t.to(self.device).contiguous()
        elif sp.env_kind != "empty":
            raise ValueError(f"env_kind {sp.env_kind!r} needs per-layout aux")
        self.join()
        cells = torch.from_numpy(layouts.pack_cells_for(sp, g.numpy()))
        old = getattr(self, "_pool", None)
        if (getattr(self, "_gen", None) is None and old is not None and old[0].shape == cells.shape and old[1].shape == a.shape
                and (old[2] is None) == (t is None)):
            # same shapes: refill the tensors in place -- sub-shards, pre-bound launchers and captured graphs stay valid
            old[0].copy_(cells); old[1].copy_(a)
            if t is not None:
                old[2].copy_(t)
            self.episode.zero_(); self.was_reset.zero_()
            return
        self._gen = None
        self._pool = (cells.to(self.device).contiguous(), a.to(self.device).contiguous(), t)
        self.episode = torch.zeros((self.batch,), dtype=torch.int32, device=self.device)
        self.was_reset = torch.zeros((self.batch,), dtype=torch.uint8, device=self.device)
        self._bound.clear()              # (the pre-bound launchers hold the old pool's pointers ...
        self._layout_version += 1        #  ... and so do sub-shards and captured graphs made before: they now refuse to run)

    def set_layout_generator(self, kind: str, layout_seed: int = 0, *, room_size: int = 0, start=(1, 1, 0),
                             max_hallway_keys: int = 1, max_keys_per_room: int = 2, staged=True, lead: int | None = None):
        """Episode starts generated ON THE DEVICE (mgx_reset_generate) instead of picked from a host-made pool: every
        finished env runs the reference's own `_gen_grid` (rejection-sampling placement with numpy-compatible draws) in a
        kernel, one lane per env.

        kind         'empty_fixed' (EmptyEnv, agents at `start` = (x, y, dir)), 'empty_random' (EmptyEnv with
                     agent_start_pos=None), 'blockedunlockpickup' (`room_size`), 'redbluedoors' (the spec's 2*size x size grid),
                     'lockedhallway' (`room_size`, `max_hallway_keys`, `max_keys_per_room`; at most 16 rooms) or 'playground'
                     (`room_size`; at most 16 rooms) -- every env class of the reference
        layout_seed  seeds every env's placement generator: Generator(PCG64(SeedSequence([layout_seed, global index])))
        `reset_done()` then regenerates every finished env; `step(auto_reset=True)` regenerates the envs whose episode ends
        with that step right after it -- in the tail of the step's own launch (mgx_step_generate), so the returned
        observation is the terminal one and the state tensors already hold the next episode's start.
        staged       how the truncation resets -- known in advance -- are kept off the step's critical path (include/mgx.h: MgxGenStage;
                     same results bit for bit in every mode, the slots are a cache):
                       True (default) / "candidates"  EVERY episode end an adoption: the generators draw from env.np_random at most
                                    once (BlockedUnlockPickup: the door row; the others never), and the placement stream does not
                                    depend on when the episode ends -- so the next episode is generated while the current one runs,
                                    once per possible value of that draw (room_size - 2 <= 4 candidates), by one generator launch
                                    (mgx_stage_generate) every lead/2 steps between two steps; a step that ends an episode -- by
                                    truncation or by success / failure -- makes the draw and adopts the matching candidate (a copy).
                                    Playground (many draws) and room sizes above 6 fall back to "between"
                       "between"    `lead` steps before an env truncates its step takes a snapshot of np_random; every
                                    lead/2 steps ONE generator launch (mgx_stage_generate) between two steps serves the pending
                                    snapshots into per-env slots; the truncating step adopts its slot (a copy).  Measured at C3 with the
                                    episodes out of phase: 9.7 us per step, against 15.6 in-launch and 16.6 unstaged (pool: 7.3)
                       "side"       the same generator launches on a stream of their own beside the steps (a parallel branch of a
                                    captured graph): measured slower than "between" (14.2 us): the fork / join of the branches costs
                                    more than the generator's few microseconds every lead/2 steps
                       "in_launch"  round 3's form: generator wavefronts appended to every step's launch, two steps ahead
                       False        no staging: every finished env is generated in the tail of its step
                     (`step()` and `capture_steps()` issue the generator launches; `rollout()` does it inside mgx_step_ex)
        lead         steps between the snapshot and the truncation (default: max_steps / 4, at most 128; 2 for "in_launch");
                     "candidates": twice the number of steps between two generator launches (default: 2/3 of max_steps, at most 512)
        """
        self._no_session("set_layout_generator")
        sp = self.spec
        self._need_wide_cells("set_layout_generator (device-side episode generation)")
        if kind == "blockedunlockpickup":
            if sp.env_kind != "blockedunlockpickup" or (sp.width, sp.height) != (2 * room_size - 1, room_size):
                raise ValueError("blockedunlockpickup generator: the spec must be a BlockedUnlockPickup grid of (2*room_size-1) x room_size")
            blank = layouts.roomgrid_blank(room_size, 1, 2)
        elif kind == "redbluedoors":
            if sp.env_kind != "redbluedoors" or sp.width != 2 * sp.height:
                raise ValueError("redbluedoors generator: the spec must be a RedBlueDoors grid of (2*size) x size")
            blank = layouts.redbluedoors_blank(sp.height)
        elif kind == "lockedhallway":
            rs = int(room_size)
            if sp.env_kind != "lockedhallway" or rs < 4 or sp.width != 3 * (rs - 1) + 1 or (sp.height - 1) % (rs - 1):
                raise ValueError("lockedhallway generator: the spec must be 3 columns of `room_size` rooms")
            blank = layouts.lockedhallway_blank(2 * ((sp.height - 1) // (rs - 1)), rs)
        elif kind == "playground":
            rs = int(room_size)
            if sp.env_kind != "empty" or rs < 4 or (sp.width - 1) % (rs - 1) or (sp.height - 1) % (rs - 1):
                raise ValueError("playground generator: the spec must be a hook-free grid of `room_size` rooms")
            blank = layouts.roomgrid_blank(rs, (sp.height - 1) // (rs - 1), (sp.width - 1) // (rs - 1))
        elif kind in ("empty_fixed", "empty_random"):
            if sp.env_kind != "empty" or sp.width != sp.height:
                raise ValueError("empty generator: the spec must be a square Empty grid")
            blank = layouts.empty_blank(sp.width)
        else:
            raise ValueError(f"unknown layout generator {kind!r}")
        self.join()
        idx = self.first_env + np.arange(self.batch)
        self._gen = {"kind": kind, "room_size": int(room_size), "start": tuple(int(v) for v in start),
                     "max_hallway_keys": int(max_hallway_keys), "max_keys_per_room": int(max_keys_per_room),
                     "blank": torch.from_numpy(layouts.pack_cells(blank).view(np.int16)).to(self.device).contiguous(),
                     "gen_state": torch.from_numpy(rnglib.layout_gen_state(layout_seed, idx).view(np.int64)).to(self.device)}
        # candidates per env (include/mgx.h: MgxGenStage.candidates): one per value of the generator's single env.np_random draw
        cand = {"blockedunlockpickup": int(room_size) - 2, "redbluedoors": 1, "lockedhallway": 1, "empty_fixed": 1,
                "empty_random": 1}.get(kind, 0)
        if staged is True:
            staged = "candidates" if 1 <= cand <= 4 else "between"
        if staged not in (False, "candidates", "between", "side", "in_launch"):
            raise ValueError(f"staged must be True / 'candidates' / 'between' / 'side' / 'in_launch' / False, got {staged!r}")
        if staged == "candidates" and not 1 <= cand <= 4:
            raise ValueError(f"staged='candidates': the {kind!r} generator draws from env.np_random more than once (or has more than "
                             f"4 possible draws): use 'between'")
        if staged == "candidates" and sp.num_agents > 1:
            B, dev = self.batch, self.device
            # (here `lead` only sets the generator launches' cadence -- every lead/2 steps, by default a third of max_steps: a launch
            # costs 15-50 us of the stream's time whatever it finds to do (profiles/r5_candidates.txt), and an episode shorter than
            # the cadence merely generates in the tail of its last step.  Callers whose episodes are short pass a smaller `lead`.)
            lead = int(lead) if lead is not None else max(4, min(512, 2 * sp.max_steps // 3))
            lead = max(2, min(lead, sp.max_steps - 1))
            self._gen["stage"] = {"lead": lead, "external": 2, "stream": None, "candidates": cand,
                                  "grid": torch.zeros((B, cand, sp.height, sp.width), dtype=torch.int16, device=dev),
                                  "agents": torch.zeros((B, cand, sp.num_agents, 8), dtype=torch.uint8, device=dev),
                                  "aux": torch.zeros((B, cand, 16), dtype=torch.uint8, device=dev) if sp.env_kind != "empty" else None,
                                  "words": torch.zeros((B, cand, 6), dtype=torch.int64, device=dev),
                                  "tag": torch.full((B, 4), -1, dtype=torch.int32, device=dev), "phase": [0]}
        elif staged and sp.num_agents > 1:       # the staging slots: a cache, not state (tag -1 = empty)
            B, dev = self.batch, self.device
            tag = torch.zeros((B, 4), dtype=torch.int32, device=dev)
            tag[:, 0] = -1
            side = staged in ("side", "between")
            lead = int(lead) if lead is not None else (max(2, min(128, sp.max_steps // 4)) if side else 2)
            if not 2 <= lead < sp.max_steps or (side and lead < 4):
                side, lead, staged = False, 2, "in_launch"      # (episodes too short to look that far ahead: the in-launch form)
            self._gen["stage"] = {"lead": lead, "external": (2 if staged == "between" else 1) if side else 0, "stream": torch.cuda.Stream(dev) if side and staged == "side" and dev.type == "cuda" else None,
                                  "grid": torch.zeros((B, sp.height, sp.width), dtype=torch.int16, device=dev),
                                  "agents": torch.zeros((B, sp.num_agents, 8), dtype=torch.uint8, device=dev),
                                  "aux": torch.zeros((B, 16), dtype=torch.uint8, device=dev) if sp.env_kind != "empty" else None,
                                  "words": torch.zeros((B, 12), dtype=torch.int64, device=dev), "tag": tag, "phase": [0]}
        self._pool = None
        self.episode = torch.zeros((self.batch,), dtype=torch.int32, device=self.device)
        self.was_reset = torch.zeros((self.batch,), dtype=torch.uint8, device=self.device)
        self._bound.clear()
        self._layout_version += 1        # (sub-shards and captured graphs made before hold the old tensors' pointers)

    def reset_done(self) -> torch.Tensor:
        """Vector-env auto-reset (build-defined; the reference leaves `if env.is_done(): env.reset()` to its caller):
        every finished env (multigrid/base.py:534-539) restarts from the layout pool, its `np_random` stream left
        running as an unseeded `reset()` does.  Returns was_reset u8[B].  Observations of the restarted envs are
        produced by the next `gen_obs()` / `step()`."""
        self._need_state()
        self.join()
        if getattr(self, "_gen", None) is not None:
            self.backend.reset_generate(self.batch, self._gen, self.cells, self.agents, self.rng, self.step_count,
                                        self.aux if self.spec.env_kind != "empty" else None, self.episode, self.was_reset)
            return self.was_reset
        if getattr(self, "_pool", None) is None:
            raise RuntimeError("call set_layout_pool() or set_layout_generator() first")
        self.backend.reset_done(self.batch, self.first_env, self._pool, self.cells, self.agents, self.step_count,
                                self.aux, self.episode, self.was_reset)
        return self.was_reset

    def outputs_to_host(self) -> dict:
        """Everything the last step / gen_obs handed back, on the host after ONE device-to-host copy (into a pinned buffer) and one
        stream synchronisation: {'reward' f64[B,A], 'obs' u8[B,A,v,v,3], 'dir', 'terminated' u8[B,A], 'truncated' u8[B], 'err'
        i32[2]} as numpy views of that buffer -- valid until the next call.  For host callers of small batches (the dict API steps one
        env: six copies of a few bytes each cost six round trips); a batch's 38 MB of observations belong on the device."""
        self.join()
        if self._out.device.type == "cpu":
            host = self._out
        else:
            if self._out_host is None:
                self._out_host = torch.empty(self._out.shape, dtype=torch.uint8, pin_memory=True)
            self._out_host.copy_(self._out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            host = self._out_host
        arr = host.numpy()
        out = {}
        for name, dt, shape, off in self._out_parts:
            npdt = {torch.float64: np.float64, torch.uint8: np.uint8, torch.int32: np.int32}[dt]
            n = int(np.prod(shape)) * np.dtype(npdt).itemsize
            out[name] = arr[off:off + n].view(npdt).reshape(shape)
        return out

    def check_errors(self):
        """Synchronises and raises ValueError if any env met an unknown action since the last check."""
        self.join()
        gb = getattr(self.backend, "_grid_bad", None)
        if gb is not None:                       # byte grids: what the step kernels counted while packing them
            unpackable, ring = (int(v) for v in gb.cpu())
            if unpackable or ring:
                gb.zero_()
                raise ValueError(f"the byte grid held {ring} outer-ring cell(s) that are not WALL = (wall, grey, 0) and {unpackable} "
                                 f"cell value(s) the packed format cannot hold (include/mgx.h)")
        count, first = (int(v) for v in self.err.cpu())
        if count:
            self._reset_err()
            raise ValueError(f"Unknown action in {count} env(s); first at local env {first} "
                             f"(valid: {int(Action.left)}..{int(Action.done)}, or {NO_ACTION} for no action)")

    def is_done(self) -> torch.Tensor:
        """multigrid/base.py:534-539 per env: bool[B]."""
        self._no_session("is_done")
        truncated = self.step_count >= self.spec.max_steps
        return truncated | (self.agents[:, :, 4] != 0).all(dim=1)

    def _no_session(self, what: str):
        """The state tensors are stale while a persistent session is open (the state lives in its launch, which writes it back when
        it ends): reading or replacing them then is a mistake, not a race to win."""
        if self._session is not None:
            raise RuntimeError(f"{what}: a persistent session is open -- the env state lives in its launch until it is closed")

    def _need_wide_cells(self, what: str):
        """Device-side generation, anything with env hooks beyond the plain step, one-hot rollouts -- and everything but the plain step
        and gen_obs on byte grids -- are compiled for the 16-bit cells only."""
        if self.spec.compact:
            raise NotImplementedError(f"{what} is not available on compact cells / byte grids (EnvSpec.cell_bytes = 1 or 3: step / "
                                      f"gen_obs / auto-reset from a layout pool / full_obs); build the env with cell_bytes = 2 for it")

    def _need_state(self):
        if not self._loaded:
            raise RuntimeError("no state loaded: call load_state() / reset() first")
        if self._session is not None:
            raise RuntimeError("a persistent session is open: the env state lives in its launch until it is closed")

    def persistent(self, max_steps: int, auto_reset: bool = False, timeout_ms: int = 2000) -> "PersistentSession":
        """A `PersistentSession` over this env: closed-loop stepping with ONE resident launch instead of one launch per step
        (include/mgx.h: mgx_step_persistent).  Same results as `step()` bit for bit.  `max_steps` bounds the launch; every wait
        inside it gives up after `timeout_ms`."""
        if not (self.spec.cell_bytes == 1 and self.spec.env_kind == "empty"):
            self._need_wide_cells("persistent stepping")                    # (compact cells: the hook-free persistent launch, round 6)
        return PersistentSession(self, max_steps, auto_reset, timeout_ms)

    # ------------------------------------------------------------------------------------------ checkpoint
    def state_dict(self) -> dict:
        """Everything a resumed run needs to continue bit-identically: the env state, and -- when auto-reset is in use --
        the layout pool, the per-env episode counters and the last `was_reset`."""
        self._no_session("state_dict")
        self.join()
        sd = {"spec": self.spec.as_dict(), "first_env": self.first_env,
              "grid": self.grid.cpu().clone(), "agents": self.agents.cpu().clone(),
              "rng": self.rng.cpu().clone(), "step_count": self.step_count.cpu().clone(),
              "aux": self.aux.cpu().clone()}
        if getattr(self, "_gen", None) is not None:
            sd["generator"] = {"kind": self._gen["kind"], "room_size": self._gen["room_size"], "start": self._gen["start"],
                               "max_hallway_keys": self._gen["max_hallway_keys"], "max_keys_per_room": self._gen["max_keys_per_room"],
                               "gen_state": self._gen["gen_state"].cpu().clone()}
            sd["episode"] = self.episode.cpu().clone()
            sd["was_reset"] = self.was_reset.cpu().clone()
        if getattr(self, "_pool", None) is not None:
            pg, pa, pt = self._pool
            sd["pool"] = {"grid": torch.from_numpy(layouts.unpack_cells_for(self.spec, pg.cpu().numpy())), "agents": pa.cpu().clone(),
                          "aux": pt.cpu().clone() if pt is not None else None}
            sd["episode"] = self.episode.cpu().clone()
            sd["was_reset"] = self.was_reset.cpu().clone()
        return sd

    def load_state_dict(self, sd: dict):
        self._no_session("load_state_dict")
        if EnvSpec.from_dict(sd["spec"]) != self.spec:
            raise ValueError("state_dict was saved for a different EnvSpec")
        if int(sd.get("first_env", self.first_env)) != self.first_env:
            raise ValueError(f"state_dict was saved for the shard starting at env {sd['first_env']}, this one starts "
                             f"at {self.first_env} (layout choice and seeds are functions of the global env index)")
        self.load_state(sd["grid"], sd["agents"], sd["rng"], sd["aux"], sd["step_count"], validate=True)
        if sd.get("generator") is not None:
            g = sd["generator"]
            self.set_layout_generator(g["kind"], 0, room_size=g["room_size"], start=g["start"],
                                      max_hallway_keys=g.get("max_hallway_keys", 1), max_keys_per_room=g.get("max_keys_per_room", 2))
            self._gen["gen_state"].copy_(g["gen_state"])
            self.episode.copy_(sd["episode"])
            self.was_reset.copy_(sd["was_reset"])
        if sd.get("pool") is not None:
            p = sd["pool"]
            self.set_layout_pool(p["grid"].numpy(), p["agents"].numpy(), p["aux"].numpy() if p["aux"] is not None else None)
            self.episode.copy_(sd["episode"])
            self.was_reset.copy_(sd["was_reset"])
