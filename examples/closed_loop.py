#!/usr/bin/env python3
"""Policy in the loop: a small torch policy reads the (one-hot) observations, samples actions, the fused kernel steps the
envs -- the loop an RL rollout worker runs.  The whole iteration (policy + mgx_step_one_hot with auto-reset) is captured
in one hipGraph, so a step costs the policy's kernels + ONE env launch and no Python.

    python examples/closed_loop.py [--workload c4] [--batch 65536] [--steps 200]
    python examples/closed_loop.py --workload c2 --persistent      (ONE resident env launch: BatchedMultiGridEnv.persistent)

Prints one JSON line: env steps/s with the policy in the loop, and the env's share of the iteration."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import workloads  # noqa: E402


def run(workload="c4", batch=None, steps=200, hidden=64, device="cuda:0", one_hot=True, sub_shards=1, iters_per_graph=4):
    """sub_shards=P > 1: the double-buffered actor loop -- the batch as P independent blocks (BatchedMultiGridEnv.split), each
    its own chain [policy(block) -> env step(block)] x iters_per_graph on its own stream, P parallel branches of one hipGraph:
    one block's policy GEMMs run while another block steps."""
    dev = torch.device(device)
    wl = workloads.make(workload, batch=batch, global_batch=max(batch or 0, workloads.GLOBAL_BATCH[workload]))
    env = wl.make_env(dev, auto_reset=True)
    B, A, v = wl.batch, wl.spec.num_agents, wl.spec.view_size
    feat = v * v * (21 if one_hot else 3)
    g = torch.Generator(device=dev); g.manual_seed(0)
    w1 = torch.randn(feat, hidden, device=dev, dtype=torch.float16, generator=g) * 0.05
    w2 = torch.randn(hidden, 7, device=dev, dtype=torch.float16, generator=g) * 0.5
    actions = torch.zeros((B, A), dtype=torch.int8, device=dev)
    obs, *_ = env.gen_obs(one_hot=True) if one_hot else env.gen_obs()
    ret = torch.zeros((B, A), dtype=torch.float64, device=dev)
    P = max(1, int(sub_shards))
    shards = env.split(P) if P > 1 else [env]
    P = len(shards)
    ranges = [getattr(sh, "_range", (0, B)) for sh in shards]

    def iteration(i):
        sh, (lo, hi) = shards[i], ranges[i]
        n = (hi - lo) * A
        x = obs[lo:hi].view(n, feat).to(torch.float16)                    # the env's output buffer, read in place
        logits = torch.relu(x @ w1) @ w2
        gumbel = -torch.log(-torch.log(torch.rand_like(logits, dtype=torch.float32).clamp_(1e-6, 1 - 1e-6)))
        actions[lo:hi].copy_((logits.float() + gumbel).argmax(dim=1).view(hi - lo, A).to(torch.int8))
        o, d, rew, term, trunc = sh.step(actions[lo:hi], auto_reset=True, one_hot=one_hot)
        ret[lo:hi].add_(rew)

    for _ in range(5):
        for i in range(P):
            iteration(i)
    n_it = iters_per_graph if P > 1 else 1
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(dev)
    others = [torch.cuda.Stream(dev) for _ in range(P - 1)]
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            for o in others:
                o.wait_stream(s)
            for i in range(P):
                with torch.cuda.stream(s if i == 0 else others[i - 1]):
                    for _ in range(n_it):
                        iteration(i)
            for o in others:
                s.wait_stream(o)
    torch.cuda.current_stream(dev).wait_stream(s)
    replays = max(1, steps // n_it)
    steps = replays * n_it
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(replays):
        graph.replay()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    env.check_errors()
    # the env launch alone, same buffers
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        env.step(actions, auto_reset=True, one_hot=one_hot)
    e1.record(); torch.cuda.synchronize(dev)
    env_ms = e0.elapsed_time(e1) / steps
    return {"workload": wl.title, "batch": B, "agents": A, "one_hot": one_hot, "policy": f"MLP {feat}-{hidden}-7 fp16, Gumbel sampling",
            "sub_shards": P, "ms_per_iteration": round(dt * 1e3 / steps, 5), "env_ms_per_step": round(env_ms, 5),
            "agent_steps_per_s": round(B * A * steps / dt), "episodes_finished": int(env.episode.sum().item()),
            "mean_return": float(ret.sum().item() / max(1, int(env.episode.sum().item())) / A)}


def run_persistent(workload="c2", batch=None, steps=200, hidden=64, device="cuda:0"):
    """The same loop over a PersistentSession (include/mgx.h: mgx_step_persistent): the env kernel is launched ONCE and stays
    resident with the env state in LDS; per iteration the policy's kernels run on the main stream, `ps.post(actions)` hands the
    actions over as tagged granules (one small kernel behind the policy), `ps.wait()` returns once every wavefront has published
    its flag behind the step's outputs.  Eager (the hand-shake's step numbers are kernel arguments), so beside it the same eager
    loop with one env launch per step.  For batches in the latency regime (c2, c3, or --batch 8192 of c4)."""
    dev = torch.device(device)
    wl = workloads.make(workload, batch=batch, global_batch=max(batch or 0, workloads.GLOBAL_BATCH[workload]))
    B, A, v = wl.batch, wl.spec.num_agents, wl.spec.view_size
    feat = v * v * 3
    g = torch.Generator(device=dev); g.manual_seed(0)
    w1 = torch.randn(feat, hidden, device=dev, dtype=torch.float16, generator=g) * 0.05
    w2 = torch.randn(hidden, 7, device=dev, dtype=torch.float16, generator=g) * 0.5

    def policy(obs, actions):
        logits = torch.relu(obs.view(B * A, feat).to(torch.float16) @ w1) @ w2
        gumbel = -torch.log(-torch.log(torch.rand_like(logits, dtype=torch.float32).clamp_(1e-6, 1 - 1e-6)))
        actions.copy_((logits.float() + gumbel).argmax(dim=1).view(B, A).to(torch.int8))

    out = {"workload": wl.title, "batch": B, "agents": A, "policy": f"MLP {feat}-{hidden}-7 fp16, Gumbel sampling, eager"}
    for mode in ("launches", "persistent"):
        env = wl.make_env(dev, auto_reset=True)
        actions = torch.zeros((B, A), dtype=torch.int8, device=dev)
        obs, _ = env.gen_obs()
        ret = torch.zeros((B, A), dtype=torch.float64, device=dev)
        for _ in range(5):                                   # (warm: allocator, code objects)
            policy(obs, actions)
        torch.cuda.synchronize(dev)
        if mode == "launches":
            t0 = time.perf_counter()
            for _ in range(steps):
                policy(obs, actions)
                _, _, rew, _, _ = env.step(actions, auto_reset=True)
                ret.add_(rew)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
        else:
            main = torch.cuda.current_stream(dev)
            with env.persistent(max_steps=steps, auto_reset=True) as ps:
                main.synchronize()          # (the STREAM: a device-wide synchronize would wait for the resident launch itself)
                t0 = time.perf_counter()
                for _ in range(steps):
                    policy(obs, actions)
                    _, _, rew, _, _ = ps.step(actions)
                    ret.add_(rew)
                main.synchronize()
                dt = time.perf_counter() - t0
            out["wavefronts_resident"] = ps.waves
        env.check_errors()
        out[mode] = {"ms_per_iteration": round(dt * 1e3 / steps, 5), "agent_steps_per_s": round(B * A * steps / dt),
                     "episodes_finished": int(env.episode.sum().item())}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--plain-obs", action="store_true", help="3-channel observations instead of one-hot")
    ap.add_argument("--sub-shards", type=int, default=1, help="double-buffered actor loop over this many blocks of the batch")
    ap.add_argument("--persistent", action="store_true",
                    help="one resident env launch (BatchedMultiGridEnv.persistent) instead of one launch per step; eager")
    a = ap.parse_args()
    if a.persistent:
        print(json.dumps(run_persistent("c2" if a.workload == "c4" and a.batch is None else a.workload, a.batch, a.steps)))
    else:
        print(json.dumps(run(a.workload, a.batch, a.steps, one_hot=not a.plain_obs, sub_shards=a.sub_shards)))
