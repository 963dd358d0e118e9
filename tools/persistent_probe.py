#!/usr/bin/env python3
"""Closed-loop stepping: ONE resident launch (mgx_step_persistent) against one launch per step, same box, same call.

    python tools/persistent_probe.py [c2|share8|c3 ...]            (product library)
    MGX_LIBMGX=multigrid_amd/lib/libmgx_spans.so python tools/persistent_probe.py c2      (+ in-kernel spans per wavefront and step)

For each workload:
  launches     T steps as hipGraph replays of mgx_step_autoreset, back to back (the bench's number: actions precomputed, a step's
               launch follows the previous one's end -- a closed loop with a policy that costs nothing and no hand-off)
  launches+p   the same with the shortest policy there can be between the steps: one kernel that copies the step's recorded actions
               into the action tensor (what a closed loop with ANY policy kernel pays: two boundaries per step)
  persistent   mgx_step_persistent fed by mgx_persistent_feed (one resident workgroup: wait for step t-1's flags, post step t's
               granules -- the shortest producer there can be), from the feeder's own trace (s_memrealtime, 10 ns):
                   step      = posted(t+1) - posted(t)
                   env       = all flags of step t seen - granules of step t posted     (the env's share of the loop)
                   producer  = granules of step t+1 posted - flags of step t seen
  spans build: per wavefront and step, [granules seen, flag stored]: the chain, the detection skew and the publish skew.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib, workloads  # noqa: E402

dev = torch.device("cuda", 0)
lib = _lib.lib()
SPANS = hasattr(lib, "mgx_debug_read_span")
if SPANS:
    lib.mgx_debug_span_launches.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    lib.mgx_debug_read_span.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]

WL = {"c2": ("c2", 4096), "share8": ("c4", 8192), "share4": ("c4", 16384), "c3": ("c3", 16384), "c3_8k": ("c3", 8192)}


def pct(x, qs=(1, 10, 50, 90, 99, 100)):
    return " ".join(f"{np.percentile(x, q):7.2f}" for q in qs)


def launches(wl, T, with_policy):
    env = wl.make_env(dev, auto_reset=True)
    B, A = wl.batch, wl.spec.num_agents
    acts = bench.random_actions(T, B, A, dev, 1234)
    cur = torch.zeros((B, A), dtype=torch.int8, device=dev)
    for t in range(50):
        env.step(acts[t % T], auto_reset=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for t in range(T):
                if with_policy:
                    cur.copy_(acts[t])
                    env.step(cur, auto_reset=True)
                else:
                    env.step(acts[t], auto_reset=True)
    torch.cuda.current_stream(dev).wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 5
    e0.record()
    for _ in range(R):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (R * T)


def persistent(wl, T):
    env = wl.make_env(dev, auto_reset=True)
    B, A = wl.batch, wl.spec.num_agents
    acts = bench.random_actions(T, B, A, dev, 1234)
    warm = bench.random_actions(50, B, A, dev, 99)
    for t in range(50):
        env.step(warm[t], auto_reset=True)
    out = {}
    for rep in range(3):                                    # (the last repetition is reported)
        if SPANS:
            lib.mgx_debug_span_reset()
        with env.persistent(max_steps=T, auto_reset=True) as ps:
            tr = ps.feed(acts, trace=True)
        torch.cuda.synchronize()
        assert ps.timeouts == 0 and ps.steps_completed == T, (ps.timeouts, ps.steps_completed)
        tr_raw = tr.cpu().numpy().astype(np.int64)
        tr = tr_raw / 100.0                                   # us
        seen, posted = tr[0::2], tr[1::2]                     # seen[t]: step t's predecessor complete; posted[t]: step t+1's granules out
        k = T // 5                                            # (skip the start: the launch ramps up, the first touches)
        out = {"step": np.diff(posted)[k:], "env": (seen[1:] - posted)[k:], "producer": (posted - seen[:-1])[k:], "waves": ps.waves}
        if SPANS:
            tab = (ctypes.c_longlong * (4 * 64))()
            nl = lib.mgx_debug_span_launches(tab, 64)
            base, nw = int(tab[4 * (nl - 1)]), ps.waves
            Tr = min(T, (1 << 18) // nw - 1)
            buf = (ctypes.c_ulonglong * (2 * nw * Tr))()
            lib.mgx_debug_read_span(buf, base, nw * Tr)
            raw = np.frombuffer(buf, dtype=np.uint64).reshape(Tr, nw, 2)
            drain = ((raw[:, :, 1] >> np.uint64(48)) & np.uint64(0xfff)).astype(np.int64)[k:Tr] / 100.0
            a = (raw & np.uint64(0xffffffffffff)).astype(np.int64) / 100.0
            a = a[k:Tr]
            p = (tr_raw[1::2] & 0xffffffffffff)[k:Tr, None] / 100.0
            seen48 = (tr_raw[0::2] & 0xffffffffffff) / 100.0
            out["spans"] = {"detect_first": (a[:, :, 0].min(axis=1) - p[:, 0]), "detect_last": (a[:, :, 0].max(axis=1) - p[:, 0]),
                            "chain": (a[:, :, 1] - a[:, :, 0]).ravel(), "chain_max": (a[:, :, 1] - a[:, :, 0]).max(axis=1),
                            "drain (vmcnt 0)": drain.ravel(),
                            "publish_last": (a[:, :, 1].max(axis=1) - p[:, 0]),
                            "seen_after_publish": seen48[1:][k:Tr] - a[:, :, 1].max(axis=1)}
    return out


def main():
    names = [a for a in sys.argv[1:] if a in WL] or ["c2", "share8", "c3"]
    T = int(os.environ.get("MGX_T", "400"))
    print(f"# library: {_lib.LIB_PATH}{' (spans build)' if SPANS else ''}; T = {T} steps per measurement; us")
    for name in names:
        base, B = WL[name]
        wl = workloads.make(base, batch=B, global_batch=max(B, workloads.GLOBAL_BATCH[base]))
        li = _lib.launch_info(wl.spec, B)
        print(f"\n== {name}: {wl.title}, {B} envs, {li['envs_per_wavefront']} envs per wavefront, fixed_shape {li['fixed_shape']}")
        l0 = launches(wl, T, False)
        l1 = launches(wl, T, True)
        print(f"launches    (graph, actions precomputed)                 {l0:7.2f} per step")
        print(f"launches+p  (graph, one copy kernel per step as policy)  {l1:7.2f} per step")
        try:
            p = persistent(wl, T)
        except _lib.MgxError as e:
            print("persistent: ", e)
            continue
        print(f"persistent  ({p['waves']} wavefronts resident; feeder = 1 workgroup per 2048 granules)   p1 p10 p50 p90 p99 max")
        for kname in ("step", "env", "producer"):
            print(f"   {kname:9s} {pct(p[kname])}    mean {p[kname].mean():7.2f}")
        if "spans" in p:
            for kname, v in p["spans"].items():
                print(f"   span {kname:19s} {pct(v)}    mean {v.mean():7.2f}")


if __name__ == "__main__":
    main()
