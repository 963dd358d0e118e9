#!/usr/bin/env python3
"""Do the sub-shard chains of a replayed step graph really overlap?  Evidence from INSIDE the kernels, independent of
bench.py's timers and of rocprofv3 (whose instrumentation serialises the chains: one launch in flight, 38.7 us per step of
C4 where the uninstrumented run takes 15.7).

Needs lib/libmgx_spans.so (`python -m multigrid_amd.build --spans`, -DMGX_SPANS=1: the product kernels + this, nothing else):
lane 0 of every wavefront stores s_memrealtime (100 MHz, one clock for the whole device) at its first and after its last
instruction into a block of records that belongs to its launch (KernelArgs::span_base, handed out in capture order).  A graph
of K steps x P chains is captured and replayed; the records of the last replay give, per graph node (chain c, step t): begin =
first wavefront's first instruction, end = last wavefront's last instruction issued.  From those: the timeline, the number of
launches in flight over time, and the step period.

    MGX_LIBMGX=multigrid_amd/lib/libmgx_spans.so MGX_WORKLOAD=c4 python tools/chain_overlap.py [batch] [P ...]  > profiles/r3_chain_overlap.txt
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

lib = _lib.lib()
if not hasattr(lib, "mgx_debug_span_reset"):
    sys.exit("needs MGX_LIBMGX=multigrid_amd/lib/libmgx_spans.so (python -m multigrid_amd.build --spans)")
lib.mgx_debug_span_launches.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
lib.mgx_debug_read_span.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
PS = [int(x) for x in sys.argv[2:]] or [1, 4]
K = int(os.environ.get("MGX_K", "48"))
SHOW = int(os.environ.get("MGX_SHOW_STEPS", "4"))

print(f"# tools/chain_overlap.py: {bench.tool_workload()} at {B} envs, graph of K={K} steps, in-kernel s_memrealtime spans (10 ns ticks)")
print("# (the spans build = the product kernels + two s_memrealtime reads and two 8-byte stores per wavefront)")
K_ASKED = K
for P in PS:
    env = bench.make_env(spec, B, dev, 0)
    nw_launch = -(-(B // P) // env.backend.launch_info(B // P)["envs_per_wavefront"])
    K = max(8, min(K_ASKED, (1 << 18) // (P * nw_launch)))          # the records of a replay must fit the 2^18-entry buffer
    acts = bench.random_actions(K, B, spec.num_agents, dev, 7)
    for t in range(K):
        env.step(acts[t], auto_reset=bench.AUTO_RESET)
    torch.cuda.synchronize()
    lib.mgx_debug_span_reset()
    graph = env.capture_steps(acts, auto_reset=bench.AUTO_RESET, sub_shards=P)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(6):                                   # warm replays; the records of the LAST replay are kept
        graph.replay()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(8):
        graph.replay()
    ev1.record()
    torch.cuda.synchronize()
    ev_us = ev0.elapsed_time(ev1) * 1e3 / (8 * K)
    tab = (ctypes.c_longlong * (4 * 4096))()
    n = lib.mgx_debug_span_launches(tab, 4096)
    L = np.frombuffer(tab, dtype=np.int64)[:4 * n].reshape(n, 4)
    nodes = []
    for base, waves, batch, first in L:
        buf = (ctypes.c_ulonglong * (2 * int(waves)))()
        if lib.mgx_debug_read_span(buf, int(base), int(waves)) != 0:
            sys.exit("mgx_debug_read_span failed (more than 2^18 wavefronts recorded?)")
        a = np.frombuffer(buf, dtype=np.uint64).reshape(int(waves), 2).astype(np.int64)
        nodes.append((int(first), int(batch), a[:, 0].min(), a[:, 1].max(), int(np.median(a[:, 1] - a[:, 0])), a))
    firsts = sorted({x[0] for x in nodes})
    chain_of = {f: i for i, f in enumerate(firsts)}
    t0 = min(x[2] for x in nodes)
    step_of, seen = [], {}
    for x in nodes:                                      # capture order = step order within a chain
        seen[x[0]] = seen.get(x[0], -1) + 1
        step_of.append(seen[x[0]])
    begins = np.array([(x[2] - t0) * 10 for x in nodes], dtype=np.int64)        # ns
    ends = np.array([(x[3] - t0) * 10 for x in nodes], dtype=np.int64)
    total = ends.max() - begins.min()
    period = total / K
    busy = float((ends - begins).sum())
    # launches in flight, sampled every 10 ns over the replay (excluding the ramp of the first and the tail of the last step)
    lo, hi = np.sort(begins)[min(P, len(begins) - 1)], np.sort(ends)[-min(P, len(ends)) - 1] if len(ends) > P else ends.max()
    grid = np.arange(lo, max(hi, lo + 10), 10)
    inflight = ((begins[None, :] <= grid[:, None]) & (grid[:, None] < ends[None, :])).sum(axis=1)
    print(f"\n== {P} chain(s) x {K} steps, {B // P} envs per launch ({nodes[0][5].shape[0]} wavefronts) ==")
    print(f"   graph replay, HIP events over 8 replays: {ev_us:.2f} us per step of the batch")
    # steady state: per chain, (begin of its last step - begin of its step K/4) / steps between -- no fork ramp, no tail
    steady = []
    for f in firsts:
        idx = [i for i, x in enumerate(nodes) if x[0] == f]
        steady.append((begins[idx[-1]] - begins[idx[K // 4]]) / (len(idx) - 1 - K // 4))
    print(f"   in-kernel: first wavefront begins at 0, last wavefront ends at {total} ns -> {period / 1e3:.2f} us per step of the batch "
          f"over the whole replay (fork ramp and tail included); steady state (per chain, begin of step {K // 4} -> begin of step {K - 1}): "
          f"{np.mean(steady) / 1e3:.2f} us per step")
    print(f"   launch duration (first wave begin -> last wave end): median {np.median(ends - begins) / 1e3:.2f} us, "
          f"min {(ends - begins).min() / 1e3:.2f}, max {(ends - begins).max() / 1e3:.2f}; median wavefront lives {np.median([x[4] for x in nodes]) * 10 / 1e3:.2f} us")
    print(f"   launches in flight: mean {busy / total:.2f} over the replay (sum of launch durations / replay span); "
          f"sampled every 10 ns in the steady part: mean {inflight.mean():.2f}, "
          + ", ".join(f"{k}: {100.0 * (inflight == k).mean():.0f} %" for k in range(0, P + 1)))
    gaps = []
    for f in firsts:
        idx = [i for i, x in enumerate(nodes) if x[0] == f]
        gaps += [begins[idx[j + 1]] - ends[idx[j]] for j in range(len(idx) - 1)]
    print(f"   gap between consecutive launches of ONE chain (end of step t -> first wave of step t+1): median {np.median(gaps):.0f} ns, "
          f"min {np.min(gaps):.0f}, max {np.max(gaps):.0f}")
    print(f"   timeline of the first {SHOW} and the last 2 steps (ns from the first wavefront of the replay):")
    for i, x in enumerate(nodes):
        if step_of[i] < SHOW or step_of[i] >= K - 2:
            print(f"     chain {chain_of[x[0]]} (envs {x[0]:6d}..{x[0] + x[1] - 1:6d}) step {step_of[i]:3d}: begin {begins[i]:8d}  end {ends[i]:8d}  "
                  f"duration {ends[i] - begins[i]:6d}")
    del graph, env
    torch.cuda.empty_cache()
