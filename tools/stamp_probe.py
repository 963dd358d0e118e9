#!/usr/bin/env python3
"""Per-phase latency of ONE wavefront of the fused step kernel from in-kernel shader-clock stamps (profiling aid).
Needs a library built with -DMGX_TIMESTAMPS=1 in place of multigrid_amd/lib/libmgx.so (tools/altlib_sweep.sh style).
Usage (GPU box): [MGX_WORKLOAD=c3 [MGX_GEN=1]] python tools/stamp_probe.py [batch ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib  # noqa: E402

lib = _lib.lib()
if not hasattr(lib, "mgx_debug_read_stamps"):
    sys.exit("library was not built with -DMGX_TIMESTAMPS=1")
lib.mgx_debug_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_longlong]
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
names = ["start", "P0", "P0end", "AR", "P1a", "P1s", "P1s_end", "P1hook", "P1d", "P2", "P3", "P4", "P5", "P5end", "P5'", "P5end'", "end"]
for B in [int(x) for x in sys.argv[1:]] or [4096]:
    env = bench.make_env(spec, B, dev, 0)
    if os.environ.get("MGX_GEN"):               # the step with episode starts generated on the device (c3; staged unless MGX_GEN=unstaged)
        env.set_layout_generator("blockedunlockpickup", layout_seed=3, room_size=6, staged=os.environ["MGX_GEN"] != "unstaged")
    acts = bench.random_actions(64, B, spec.num_agents, dev, 7)
    li = env.backend.launch_info(B)
    nw = (B + li["envs_per_wavefront"] - 1) // li["envs_per_wavefront"]
    for t in range(20):
        env.step(acts[t], auto_reset=bench.AUTO_RESET)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    print(f"B={B} launch {li}")
    for wave in (0, nw // 4, nw // 2, (3 * nw) // 4, nw - 1):
        lib.mgx_debug_read_stamps(buf, wave)          # select the wave (and clear)
        acc = None
        reps = 20
        for r in range(reps):
            if os.environ.get("MGX_GEN_ALL"):       # every env truncates with this step: every wavefront generates in its tail
                env.step_count.fill_(env.spec.max_steps - 1)
            env.step(acts[20 + r], auto_reset=bench.AUTO_RESET)
            torch.cuda.synchronize()
            lib.mgx_debug_read_stamps(buf, wave)
            st = [int(x) for x in buf if x]
            d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
            if acc is None:
                acc, tot_all, all_d = d, [], []
            acc = [min(x, y) for x, y in zip(acc, d)]     # min over launches: least disturbed
            tot_all.append(st[-1] - st[0])
            all_d.append(d)
        # (the marker sequence after "P4" is one ("P5", "P5end") pair per staging round of 16 slots, then "end")
        labels = (names[:12] + ["P5", "P5end"] * 8)[:len(acc) - 0]
        labels = labels[:len(acc)] + ["?"] * (len(acc) - len(labels))
        tot = sum(acc)
        print(f"  wave {wave}: sum of per-phase minima {tot} clk; whole wave: min {min(tot_all)} median {sorted(tot_all)[len(tot_all) // 2]} max {max(tot_all)} clk  " + "  ".join(f"{labels[i]}>{acc[i]}" for i in range(len(acc))))
        med = [sorted(x[i] for x in all_d if len(x) == len(acc))[len(all_d) // 2] for i in range(len(acc))]
        print("      medians over launches: " + "  ".join(f"{labels[i]}>{med[i]}" for i in range(len(acc))))
    del env
