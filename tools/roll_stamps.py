#!/usr/bin/env python3
"""Per-phase latency of ONE wavefront of the ROLLOUT kernel (mgx_rollout_autoreset: the envs' state resident in LDS) from in-kernel
shader-clock stamps: where does a wavefront's step go?  Needs the timestamps build:

    python -c "from multigrid_amd import build; build.build_lib(False, False, build.LIB_TS, ('MGX_DEBUG_KNOBS=1','MGX_TIMESTAMPS=1','MGX_SINGLE_TU=1','MGX_NO_BIG_PERSIST=1','MGX_ONLY_V=7'))"
    MGX_LIBMGX=multigrid_amd/lib/libmgx_ts.so [MGX_RESIDENT_SLICES=1] python tools/roll_stamps.py [batch ...]

The kernel stamps `start, P0, P0end` once and then, per step, `AR P1a P1s P1s_end P1hook P1d P2 P3 P4` and one `(P5, P5end)` pair per
staging round (P4 of the later rounds lies between a P5end and the next P5); 64 stamps = the first three steps of the wavefront.
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib, workloads  # noqa: E402

lib = _lib.lib()
if not hasattr(lib, "mgx_debug_read_stamps"):
    sys.exit("library was not built with -DMGX_TIMESTAMPS=1")
lib.mgx_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_longlong]
dev = torch.device("cuda", 0)
name = os.environ.get("MGX_WORKLOAD", "c4")
for B in [int(x) for x in sys.argv[1:]] or [49152]:
    wl = workloads.make(name, batch=B, global_batch=max(B, workloads.GLOBAL_BATCH[name]))
    env = wl.make_env(dev, auto_reset=True)
    A = wl.spec.num_agents
    T = 4
    acts = bench.random_actions(64, B, A, dev, 7)
    for t in range(30):
        env.step(acts[t], auto_reset=True)
    torch.cuda.synchronize()
    info = _lib.launch_info(wl.spec, B, roll=True)
    nw = info["wavefronts"]
    per_step = 9 + 2 * (info["envs_per_slice"] * A // (8 if info["slices"] > 1 else 16))
    print(f"B={B} rollout geometry {info}; {per_step} stamps per (step, slice)")
    buf = (ctypes.c_ulonglong * 64)()
    step_names = ["AR", "P1a", "P1s", "P1s_end", "P1hook", "P1d", "P2", "P3", "P4"]
    for wave in (0, nw // 2, nw - 1):
        lib.mgx_debug_read_stamps(buf, wave)
        rows = []
        for r in range(12):
            env.rollout(acts[30 + r:30 + r + T].contiguous(), auto_reset=True)
            torch.cuda.synchronize()
            lib.mgx_debug_read_stamps(buf, wave)
            st = [int(x) for x in buf if x]
            rows.append(np.diff(np.asarray(st, dtype=np.int64)))
        n = min(len(x) for x in rows)
        d = np.median(np.stack([x[:n] for x in rows]), axis=0)
        labels = ["start", "P0", "P0end"]
        while len(labels) < n + 1:
            labels += step_names + ["P5", "P5end"] * ((per_step - 9) // 2)
        print(f"  wave {wave}: medians over 12 launches, shader clocks (100 MHz realtime is NOT used: s_memtime cycles)")
        k = 0
        print("    prologue: " + "  ".join(f"{labels[i]}>{int(d[i])}" for i in range(3)))
        k = 3
        s_idx = 0
        while k + per_step <= n:
            seg = d[k:k + per_step]
            p45 = seg[9:]
            print(f"    step-slice {s_idx}: total {int(seg.sum())}  " + "  ".join(f"{step_names[i]}>{int(seg[i])}" for i in range(9))
                  + f"  | P5 rounds: {[int(x) for x in p45[0::2]]}  P5end->next (P4 of the next round): {[int(x) for x in p45[1::2]]}")
            k += per_step
            s_idx += 1
    del env
