import sys, zlib, numpy as np, torch
sys.path.insert(0, '.')
from multigrid_amd import BatchedMultiGridEnv, EnvSpec
from oracle import binding as ob
from tests import util
spec = EnvSpec(16, 16, 4, 7, max_steps=1024)
for B in (1, 2, 4, 8, 64, 4096):
    st = util.random_state(spec, B, seed=5)
    env = BatchedMultiGridEnv(spec, B, 'cuda:0')
    env.load_state(st["grid"], st["agents"], st["rng"], st["target"], st["step_count"])
    o_ref, d_ref = ob.gen_obs_batch(spec.as_dict(), st["grid"], st["agents"], nthreads=8)
    obs, dirs = env.gen_obs()
    o = obs.cpu().numpy()
    bad = np.argwhere((o != o_ref).any(-1))
    print("B", B, "launch", env.backend.launch_info(B), "bad cells", len(bad))
    for b, a, i, j in bad[:12]:
        print("  env", b, "agent", a, "i,j", i, j, "got", o[b, a, i, j], "want", o_ref[b, a, i, j], "agent row", st["agents"][b, a])
    if len(bad):
        import collections
        print("  by agent", collections.Counter(bad[:, 1].tolist()), "by env%4", collections.Counter((bad[:, 0] % 4).tolist()))
        print("  by (i,j)", collections.Counter(map(tuple, bad[:, 2:].tolist())).most_common(8))
