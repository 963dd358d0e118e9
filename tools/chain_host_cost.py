#!/usr/bin/env python3
"""Where does the host time of an eager sub-shard step go (32 us per step, host-bound, whatever the chain count)?
Times the pieces of BatchedMultiGridEnv.step(sub_shards=P) separately, C4 at 65536 envs:
    event record on the current stream | the ctypes call (mgx_step_chains: P x (hipStreamWaitEvent + launch)) | P x Tensor.record_stream
"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import workloads  # noqa: E402

dev = torch.device("cuda", 0)
wl = workloads.make("c4")
for P in (2, 4):
    env = wl.make_env(dev, auto_reset=True)
    acts = list(bench.random_actions(64, wl.batch, 4, dev, 7))
    for t in range(100):
        env.step(acts[t & 63], auto_reset=True, sub_shards=P)
    env.join(); torch.cuda.synchronize()
    N = 2000
    t0 = time.perf_counter()
    for t in range(N):
        env.step(acts[t & 63], auto_reset=True, sub_shards=P)
    t_all = (time.perf_counter() - t0) / N
    env.join(); torch.cuda.synchronize()
    key = (True, False, False, P)
    fast = env._bound[key]
    ev = env._fork_event
    cur = torch.cuda.current_stream(dev)
    t0 = time.perf_counter()
    for t in range(N):
        ev.record(cur)
    t_ev = (time.perf_counter() - t0) / N
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(N):
        fast(acts[t & 63], ev.cuda_event, None)
    t_call = (time.perf_counter() - t0) / N
    env._chains_pending, env._chains_P = True, P
    env.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(N):
        for st in env._chain_streams[:P]:
            acts[t & 63].record_stream(st)
    t_rs = (time.perf_counter() - t0) / N
    # the one-launch form for comparison
    t0 = time.perf_counter()
    for t in range(N):
        env.step(acts[t & 63], auto_reset=True)
    t_one = (time.perf_counter() - t0) / N
    torch.cuda.synchronize()
    print(f"P={P}: step(sub_shards=P) host {t_all * 1e6:.1f} us/call = event record {t_ev * 1e6:.1f} + mgx_step_chains call {t_call * 1e6:.1f} "
          f"+ {P} x record_stream {t_rs * 1e6:.1f} (+ python);  step() one launch: {t_one * 1e6:.1f} us/call host")
    del env
