#!/usr/bin/env python3
"""Host cost per step call (us), C2 shape, 4096 envs: where the ~7.7 us of a Python-level env.step go (profiling aid)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib, ops, workloads  # noqa: E402

dev = torch.device("cuda", 0)
wl = workloads.make("c2")
env = wl.make_env(dev, auto_reset=True)
acts = bench.random_actions(64, wl.batch, wl.spec.num_agents, dev, 3)
N = 3000


def timeit(name, fn):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name:60s} host {1e6 * (t1 - t0) / N:6.2f} us/call   wall {1e6 * (t2 - t0) / N:6.2f} us/call", flush=True)


i = [0]
def a():
    env.step(acts[i[0] & 63], auto_reset=True); i[0] += 1
timeit("env.step (Python: checks + bound mgx_step_ex via ctypes)", a)

bound = env._bound[(True, False, False, 1)]
a0 = acts[0]
timeit("bound launcher only (ctypes mgx_step_ex + stream lookup)", lambda: bound(a0))

sa, keep = env.backend.step_args(env.cells, env.agents, env.rng, env.step_count, None, env.err, env.obs, env.dir, env.reward,
                                 env.terminated, env.truncated, auto_reset=env._auto_reset_args(True, env.was_reset))
sa.actions = a0.data_ptr()
fn = _lib.lib().mgx_step_ex
spec_ref, args_ref, B = C.byref(env.backend.sc), C.byref(sa), wl.batch
stream = torch.cuda.current_stream(dev).cuda_stream
timeit("raw ctypes call, stream handle cached", lambda: fn(spec_ref, B, args_ref, stream))
timeit("torch.cuda.current_stream(dev).cuda_stream alone", lambda: torch.cuda.current_stream(dev).cuda_stream)
ints = ops.spec_to_ints(wl.spec)
timeit("torch.ops.mgx.step (compiled op: allocates 5 outputs)",
       lambda: torch.ops.mgx.step(env.cells, env.agents, env.rng, env.step_count, a0, None, env.err, ints))
step_out = torch.ops.mgx.step_out
timeit("torch.ops.mgx.step_out (compiled op: the caller's outputs)",
       lambda: step_out(env.cells, env.agents, env.rng, env.step_count, a0, None, env.err, ints, env.obs, env.dir, env.reward,
                        env.terminated, env.truncated))
pg, pa, _ = env._pool
ar_out = torch.ops.mgx.step_autoreset_out
timeit("torch.ops.mgx.step_autoreset_out (+ layout pool, was_reset)",
       lambda: ar_out(env.cells, env.agents, env.rng, env.step_count, a0, None, env.err, pg, pa, None, env.episode, 0, ints, env.obs,
                      env.dir, env.reward, env.terminated, env.truncated, env.was_reset))
h = torch.ops.mgx.bind_step(env.cells, env.agents, env.rng, env.step_count, None, env.err, ints, env.obs, env.dir, env.reward,
                            env.terminated, env.truncated, pg, pa, None, env.episode, 0, env.was_reset, False)
step_bound = torch.ops.mgx.step_bound
timeit("torch.ops.mgx.step_bound (handle, actions): bound once, with auto-reset", lambda: step_bound(h, a0))
torch.ops.mgx.unbind_step(h)
x = torch.zeros(16, device=dev)
timeit("x.add_(1) (a torch elementwise launch, for scale)", lambda: x.add_(1))
