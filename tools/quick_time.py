#!/usr/bin/env python3
"""Fused step / gen_obs kernel time at one or more batches with the library currently in place (profiling aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
if os.environ.get("MGX_SKIP"):          # tools' build only (MGX_LIBMGX=lib/libmgx_dbg.so): mgx_debug_skip_phases bit mask
    from multigrid_amd import _lib
    _lib.lib().mgx_debug_skip_phases(int(os.environ["MGX_SKIP"]))
out = []
for B in [int(x) for x in sys.argv[1:]] or [4096, 1 << 20]:
    env = bench.make_env(spec, B, dev, 0)
    acts = bench.random_actions(4, B, spec.num_agents, dev, 7)
    i = [0]
    def step():
        env.step(acts[i[0] & 3]); i[0] += 1
    t = bench.kernel_time_ms(step, 40, dev) * 1e3
    o = bench.kernel_time_ms(env.gen_obs, 40, dev) * 1e3
    out.append(f"B={B}: step {t:.1f} us gen_obs {o:.1f} us")
    del env
print(" | ".join(out))
