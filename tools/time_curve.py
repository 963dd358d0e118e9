import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
dev = torch.device("cuda", 0)
spec = bench.workload_spec()
B = 1 << 20
env = bench.make_env(spec, B, dev, 0)
acts = bench.random_actions(16, B, spec.num_agents, dev, 7)
i = [0]
def step():
    env.step(acts[i[0] & 15], auto_reset=True); i[0] += 1
out = []
for k in range(16):
    out.append(round(bench.kernel_time_ms(step, 50, dev, warm=0) * 1e3, 1))
print("step us per 50-step window:", out)
out = []
for k in range(4):
    out.append(round(bench.kernel_time_ms(env.gen_obs, 50, dev, warm=0) * 1e3, 1))
print("gen_obs:", out, "episodes", int(env.episode.sum()))
