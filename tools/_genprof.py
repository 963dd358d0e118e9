import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
for steady in (False, True):
    r = bench.config_point("c3", dev, 256, 50, device_generated=True, steady=steady)
    print("steady" if steady else "in-phase", r["ms_per_step"], flush=True)
r = bench.config_point("c3", dev, 256, 50)
print("pool", r["ms_per_step"], flush=True)
