#!/usr/bin/env python3
"""Spans of a step launch WITH device generation (MGX_LIBMGX = the spans build): the step's wavefronts by what their tail did
(adopted a staged episode / generated in the tail / nothing) and the generator wavefronts.  Usage: python tools/gen_span_probe.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib, workloads
lib = _lib.lib()
lib.mgx_debug_span_launches.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
lib.mgx_debug_read_span.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
lib.mgx_debug_read_span_flags.argtypes = [ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda", 0)
B = 16384
wl = workloads.make("c3", batch=B, first_env=0, global_batch=B)
for staged, out_of_phase in (("pool", True), (True, True), (True, False), (False, True)):
    env = wl.make_env(dev, auto_reset=True)
    if staged != "pool":                                    # ("pool": the fused auto-reset from the host-made layout pool, for scale)
        env.set_layout_generator("blockedunlockpickup", layout_seed=3, room_size=6, staged=staged)
    acts = bench.random_actions(64, B, 2, dev, 7)
    # episodes out of phase: a steady ~28 truncations per step (B / max_steps) instead of all of them in one
    if out_of_phase:
      env.step_count.copy_(torch.randint(0, env.spec.max_steps - 3, (B,), device=dev, dtype=torch.int32,
                                       generator=torch.Generator(device=dev).manual_seed(1)))
    for t in range(40):
        env.step(acts[t % 64], auto_reset=True)
    torch.cuda.synchronize()
    r_ = bench.measure_steps(env, 20, 5, "graph", lambda: None, 7)
    print(f"staged={staged} {'out of phase' if out_of_phase else 'in phase (no env ends in these launches)'}: graph replay {r_['event_ms'] * 1e3 / r_['timed_steps']:.2f} us per step, episodes finished {int(env.episode.sum())}")
    D, F = [], []
    for r in range(12):
        lib.mgx_debug_span_reset()
        env.step(acts[(40 + r) % 64], auto_reset=True)
        torch.cuda.synchronize()
        tab = (ctypes.c_longlong * (4 * 8))()
        nl = lib.mgx_debug_span_launches(tab, 8)
        base, nw = int(tab[4 * (nl - 1)]), int(tab[4 * (nl - 1) + 1])
        buf = (ctypes.c_ulonglong * (2 * nw))(); fl = (ctypes.c_ubyte * nw)()
        lib.mgx_debug_read_span(buf, base, nw); lib.mgx_debug_read_span_flags(fl, base, nw)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 2).astype(np.int64); f = np.frombuffer(fl, dtype=np.uint8).copy()
        ok = a[:, 0] > 0
        t0 = a[ok, 0].min()
        D.append(np.stack([(a[ok, 0] - t0) * 10, (a[ok, 1] - t0) * 10], 1)); F.append(f[ok])
        if r == 0:
            print(f"staged={staged}: {nw} wavefront records in the launch ({int(ok.sum())} written), launch ends at +{int(D[-1][:,1].max())} ns")
    D, F = np.concatenate(D), np.concatenate(F)
    d = D[:, 1] - D[:, 0]
    pc = lambda x: " ".join(f"{int(np.percentile(x, q)):6d}" for q in (1, 50, 90, 99, 100)) if len(x) else "-"
    print(f"   wavefronts begin (ns after the launch's first) p1 p50 p90 p99 max {pc(D[:, 0])}")
    for name, m in (("step waves, nothing in the tail", (F & 15) == 0), ("adopted a staged episode (1)", (F & 1) != 0), ("generated in the tail (2)", (F & 2) != 0),
                    ("left a snapshot (4), nothing else", (F & 7) == 4),
                    ("generator waves that generated (8)", (F & 8) != 0)):
        print(f"   {name:36s} {int(m.sum()):6d} waves: duration p1 p50 p90 p99 max {pc(d[m])}   end {pc(D[m, 1])}")
