import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
for K in (64, 256, 512, 1000, 2000):
    for P in (4, 2):
        env = bench.make_env(spec, 65536, dev, 0)
        m = bench.measure_steps(env, K, 20, "graph", lambda: None, seed=7, min_region_ms=60.0, sub_shards=P)
        print(f"block={m['block']} P={P}: {m['event_ms'] / m['timed_steps'] * 1e3:.2f} us per step, repeats {m['repeats']}", flush=True)
        del env
