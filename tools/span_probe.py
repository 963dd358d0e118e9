#!/usr/bin/env python3
"""When do the wavefronts of one fused-step launch begin and end (s_memrealtime, 10 ns ticks)?  Needs
MGX_LIBMGX=multigrid_amd/lib/libmgx_ts.so (`python -m multigrid_amd.build --timestamps`).  Shows the launch ramp, the per-wave
duration and the tail.  (Several chains of launches in one graph: tools/chain_overlap.py.)"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from multigrid_amd import _lib
lib = _lib.lib()
lib.mgx_debug_span_launches.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
lib.mgx_debug_read_span.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int, ctypes.c_int]
if hasattr(lib, "mgx_debug_read_span_flags"):
    lib.mgx_debug_read_span_flags.argtypes = [ctypes.POINTER(ctypes.c_ubyte), ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
if os.environ.get("MGX_WPB"):
    lib.mgx_debug_set_waves_per_workgroup(int(os.environ["MGX_WPB"]))
if os.environ.get("MGX_G"):
    lib.mgx_debug_set_envs_per_wavefront(int(os.environ["MGX_G"]))
if os.environ.get("MGX_SKIP"):
    lib.mgx_debug_skip_phases(int(os.environ["MGX_SKIP"]))
for B in [int(x) for x in sys.argv[1:]] or [4096]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(64, B, spec.num_agents, dev, 7)
    li = env.backend.launch_info(B)
    nw = (B + li["envs_per_wavefront"] - 1) // li["envs_per_wavefront"]
    for t in range(30):
        env.step(acts[t], auto_reset=bench.AUTO_RESET)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (2 * nw))()
    rows = []
    graph = None
    if os.environ.get("MGX_GRAPH"):                        # the last launch of a replayed graph of 20 steps
        lib.mgx_debug_span_reset()
        graph = torch.cuda.CUDAGraph()
        s_ = torch.cuda.Stream(dev); s_.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s_):
            with torch.cuda.graph(graph, stream=s_):
                for t in range(int(os.environ.get("MGX_GRAPH_STEPS", "20"))):   # (2^18 wave records in all: fewer steps for big launches)
                    env.step(acts[t], auto_reset=bench.AUTO_RESET)
        torch.cuda.current_stream(dev).wait_stream(s_)
    for r in range(10):
        if graph is not None:
            graph.replay()
        else:
            lib.mgx_debug_span_reset()
            env.step(acts[30 + r], auto_reset=bench.AUTO_RESET)
        torch.cuda.synchronize()
        tab = (ctypes.c_longlong * (4 * 64))()
        nl = lib.mgx_debug_span_launches(tab, 64)
        lib.mgx_debug_read_span(buf, int(tab[4 * (nl - 1)]), nw)
        a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 2).astype(np.int64)
        t0 = a[:, 0].min()
        b, e = (a[:, 0] - t0) * 10, (a[:, 1] - t0) * 10          # ns
        if hasattr(lib, "mgx_debug_read_span_flags"):  # which rare paths the waves took, and what that cost them
            fl = (ctypes.c_ubyte * nw)()
            lib.mgx_debug_read_span_flags(fl, int(tab[4 * (nl - 1)]), nw)
            acc_f = np.concatenate([acc_f, np.frombuffer(fl, dtype=np.uint8).copy()]) if r else np.frombuffer(fl, dtype=np.uint8).copy()
            acc_d = np.concatenate([acc_d, e - b]) if r else (e - b)
        if r == 9 and hasattr(lib, "mgx_debug_read_span_flags"):  # (over the last launch of all ten runs)
            fl, d = acc_f, acc_d
            nw_all = len(d)
            pc = lambda x: " ".join(f"{int(np.percentile(x, q)):6d}" for q in (1, 10, 25, 50, 75, 90, 99, 100))
            print(f"   wave durations (ns) p1 p10 p25 p50 p75 p90 p99 max, all {nw_all} waves of 10 launches: {pc(d)}")
            for bit, name in ((1, "auto-reset"), (2, "sequential fallback"), (4, "success/failure events"), (8, "cell writes")):
                m_ = (fl & bit) != 0
                if m_.any():
                    print(f"     took {name:24s} {int(m_.sum()):5d} waves: {pc(d[m_])}")
            m_ = fl == 0
            if m_.any():
                print(f"     none of those            {int(m_.sum()):5d} waves: {pc(d[m_])}")
        rows.append((b.max(), np.median(e - b), (e - b).max(), e.max(), np.percentile(e, 50), np.percentile(e, 99)))
        wpb_ = li["threads_per_workgroup"] // 64
        wg = np.arange(nw) // wpb_
        print(f"   launch {r}: begin by XCD:", " ".join(f"{int(np.median(b[wg % 8 == k])):5d}" for k in range(8)),
              "| end:", " ".join(f"{int(np.median(e[wg % 8 == k])):5d}" for k in range(8)))
    wpb_ = li["threads_per_workgroup"] // 64
    wg = np.arange(nw) // wpb_
    print("   begin (ns, last launch) by workgroup id mod 8 (= XCD):", " ".join(f"{int(np.median(b[wg % 8 == k]))}" for k in range(8)),
          "| end:", " ".join(f"{int(np.median(e[wg % 8 == k]))}" for k in range(8)))
    print("   begin by position of the workgroup within its XCD's share (quartiles):",
          " ".join(f"{int(np.median(b[(wg // 8) * 4 // max(1, (wg.max() // 8 + 1)) == q]))}" for q in range(4)))
    m = np.median(np.array(rows), axis=0)
    print(f"B={B} waves={nw}: last wave begins at +{m[0]:.0f} ns; wave duration median {m[1]:.0f} / max {m[2]:.0f} ns; "
          f"waves end: median +{m[4]:.0f}, p99 +{m[5]:.0f}, last +{m[3]:.0f} ns (from the first wave's first instruction)")
