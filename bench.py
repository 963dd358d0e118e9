#!/usr/bin/env python3
"""bench.py -- throughput of the fused HIP step (action apply + gen_obs) on synthetic random-action rollouts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_per_gpu] [--mode graph|eager]

One "step" = one `MultiGridEnv.step` over the whole per-GPU batch = one launch of the fused kernel.
Workload at N=1 = BASELINE.json configs[1]: MultiGrid-Empty-16x16-v0, agents=4, view_size=7, batch=4096 envs.
For N>1 (launched by `python -m torch.distributed.run --nproc-per-node N ...`, one rank per GPU) every rank owns
an independent shard of `--batch` envs (weak scaling); the data path has NO collective -- envs never interact
 -- and torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline            the fused kernel on the timed workload: algorithmic bytes per launch (SURVEY.md 8d:
                      339 B per agent-step for this shape) / average launch duration from HIP events over the
                      timed region on the launch stream
  roofline_large      same kernel at a working set >> the 256 MiB Infinity Cache (HBM-resident regime)
  gen_obs_large       the observation-only kernel (311 B per agent-view) at the same large working set
  cpu_baseline        the CPU oracle (a port of the reference algorithm; the Python reference cannot travel to
                      the GPU box) timed on this host's cores on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from multigrid_amd import BatchedMultiGridEnv, EnvSpec, layouts  # noqa: E402
from multigrid_amd.sharding import shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
AUTO_RESET = True              # finished envs (all terminated / truncated) restart; fused into the step launch


def workload_spec() -> EnvSpec:
    # MultiGrid-Empty-16x16-v0: multigrid/envs/__init__.py:46, empty.py:145 (max_steps = 4*size^2)
    if os.environ.get("MGX_WORKLOAD", "c2") == "c5":      # profiling tools only: BASELINE.json configs[4] (64x64, 16 agents, v=9)
        return EnvSpec(width=64, height=64, num_agents=16, view_size=9, max_steps=4 * 64 * 64, env_kind="empty")
    return EnvSpec(width=16, height=16, num_agents=4, view_size=7, max_steps=4 * 16 * 16, env_kind="empty")


def make_env(spec, batch, device, first_env, seed=1234):
    env = BatchedMultiGridEnv(spec, batch, device, first_env=first_env)
    grid, agents = layouts.empty_layout(spec.width, spec.num_agents)      # agents at (1,1) facing right
    if spec.width == 64:                                                  # C5: random interior starts (host default_rng(5))
        r = np.random.default_rng(5)
        agents = np.broadcast_to(agents, (batch,) + agents.shape).copy()
        agents[..., 2] = r.integers(1, 63, size=agents.shape[:2]); agents[..., 3] = r.integers(1, 63, size=agents.shape[:2])
        agents[..., 1] = r.integers(0, 4, size=agents.shape[:2])
        grid = np.broadcast_to(grid, (batch,) + grid.shape)
    env.load_state(grid, agents)
    env.seed_synthetic(seed)
    if AUTO_RESET:                                                        # the reference's reset() of this env class is one
        g1, a1 = layouts.empty_layout(spec.width, spec.num_agents)         # fixed layout (empty.py:151-170): a pool of K = 1
        env.set_layout_pool(g1[None], a1[None])
    return env


def random_actions(steps, batch, agents, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 7, (steps, batch, agents), dtype=torch.int8, device=device, generator=g)


def timed_rollout(env, actions, mode, dist_barrier):
    """Time exactly len(actions) steps.  Returns (wall seconds incl. sync, HIP-event ms over the region)."""
    K = actions.shape[0]
    stream = torch.cuda.current_stream(env.device)
    graph = None
    if mode == "graph":
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(env.device)
        s.wait_stream(stream)
        with torch.cuda.stream(s):
            with torch.cuda.graph(graph, stream=s):
                for t in range(K):
                    env.step(actions[t], auto_reset=AUTO_RESET)
        stream.wait_stream(s)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist_barrier()
    torch.cuda.synchronize(env.device)
    t0 = time.perf_counter()
    ev0.record(stream)
    if graph is not None:
        graph.replay()
    else:
        for t in range(K):
            env.step(actions[t], auto_reset=AUTO_RESET)
    ev1.record(stream)
    torch.cuda.synchronize(env.device)
    t1 = time.perf_counter()
    dist_barrier()
    return t1 - t0, ev0.elapsed_time(ev1)


def kernel_time_ms(fn, iters, device):
    """Average duration of `fn`'s single kernel launch: `iters` back-to-back launches between two HIP events."""
    stream = torch.cuda.current_stream(device)
    for _ in range(max(3, iters // 2)):          # also lets the clocks settle
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    ev0.record(stream)
    for _ in range(iters):
        fn()
    ev1.record(stream)
    torch.cuda.synchronize(device)
    return ev0.elapsed_time(ev1) / iters


def pmc_traffic(kernel: str, batch: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), or None if this
    (kernel, batch) was not profiled.  bench.py cannot collect PMC counters on itself."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            return json.load(fh)[kernel][str(batch)]["bytes"]
    except (OSError, KeyError, ValueError):
        return None


def roofline(alg_bytes_per_launch, ms, traffic=None):
    achieved = alg_bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic}


def large_batch_points(spec, device, large_batch):
    env = make_env(spec, large_batch, device, 0)
    acts = random_actions(4, large_batch, spec.num_agents, device, 99)
    i = [0]

    def step():
        env.step(acts[i[0] & 3], auto_reset=AUTO_RESET); i[0] += 1
    ms_step = kernel_time_ms(step, 60, device)
    ms_obs = kernel_time_ms(env.gen_obs, 60, device)
    n = large_batch * spec.num_agents
    r_step = roofline(n * spec.bytes_step(), ms_step, pmc_traffic("step", large_batch))
    r_step.update(batch=large_batch, kernel="mgx_fused_kernel<7,step>", ms_per_launch=round(ms_step, 4),
                  agent_steps_per_s=round(n / (ms_step * 1e-3)))
    r_obs = roofline(n * spec.bytes_gen_obs(), ms_obs, pmc_traffic("gen_obs", large_batch))
    r_obs.update(batch=large_batch, kernel="mgx_fused_kernel<7,gen_obs>", ms_per_launch=round(ms_obs, 4),
                 agent_views_per_s=round(n / (ms_obs * 1e-3)))
    del env
    torch.cuda.empty_cache()
    return r_step, r_obs


def aux_kernel_points(device, batch=1 << 20):
    """Bandwidth of the kernels either side of the fused step at an HBM-resident size."""
    spec = workload_spec()
    A, V, H, W = spec.num_agents, spec.view_size, spec.height, spec.width
    env = make_env(spec, batch, device, 0)
    acts = random_actions(4, batch, A, device, 7)
    for t in range(4):
        env.step(acts[t])
    out = {}
    # one-hot of the step's observations: 3 B in, 21 B out per view cell (wrappers.py:158-190)
    cells = batch * A * V * V
    t = kernel_time_ms(env.one_hot_obs, 30, device)
    by = cells * (3 + 21)
    out["one_hot"] = {"batch": batch, "ms_per_launch": round(t, 5), "algorithmic_bytes": by,
                      "achieved_GBs": round(by / t / 1e6, 1), "frac": round(by / t / 1e6 / HBM_PEAK_GBS, 4)}
    # fully observable encode: grid in, transposed grid out, agent rows in
    t = kernel_time_ms(env.full_obs, 30, device)
    by = batch * (2 * H * W * 3 + A * 8)
    out["full_obs"] = {"batch": batch, "ms_per_launch": round(t, 5), "algorithmic_bytes": by,
                       "achieved_GBs": round(by / t / 1e6, 1), "frac": round(by / t / 1e6 / HBM_PEAK_GBS, 4)}
    # auto-reset with every env done: agent rows + step counts in, layout out
    grid, agents = layouts.empty_layout(W, A)
    K = 64
    env.set_layout_pool(np.broadcast_to(grid, (K,) + grid.shape).copy(), np.broadcast_to(agents, (K,) + agents.shape).copy())

    def reset_all():
        env.step_count.fill_(spec.max_steps)          # every env truncated -> every env is reset
        env.reset_done()
    def fill_only():
        env.step_count.fill_(spec.max_steps)
    t = kernel_time_ms(reset_all, 30, device) - kernel_time_ms(fill_only, 30, device)
    by = batch * (H * W * 3 + 2 * A * 8 + 4 + 4 + 4 + 1)
    out["reset_done_all"] = {"batch": batch, "ms_per_launch": round(t, 5), "algorithmic_bytes": by,
                             "achieved_GBs": round(by / t / 1e6, 1), "frac": round(by / t / 1e6 / HBM_PEAK_GBS, 4)}
    env.step_count.zero_()
    t = kernel_time_ms(env.reset_done, 30, device)     # nobody done: the scan only
    by = batch * (A * 8 + 4 + 1)
    out["reset_done_none"] = {"batch": batch, "ms_per_launch": round(t, 5), "algorithmic_bytes": by,
                              "achieved_GBs": round(by / t / 1e6, 1), "frac": round(by / t / 1e6 / HBM_PEAK_GBS, 4)}
    return out



def rollout_point(spec, batch, device, steps, first_env, seed):
    """The same K steps as ONE mgx_rollout launch (env state stays in LDS between steps).  Open-loop actions only."""
    env = make_env(spec, batch, device, first_env)
    acts = random_actions(steps, batch, spec.num_agents, device, seed)
    out = env.rollout(acts[:2].contiguous(), auto_reset=AUTO_RESET)   # warm-up + allocation pattern
    A, v = spec.num_agents, spec.view_size
    out = {"obs": torch.empty((steps, batch, A, v, v, 3), dtype=torch.uint8, device=device),
           "dir": torch.empty((steps, batch, A), dtype=torch.uint8, device=device),
           "reward": torch.empty((steps, batch, A), dtype=torch.float64, device=device),
           "terminated": torch.empty((steps, batch, A), dtype=torch.uint8, device=device),
           "truncated": torch.empty((steps, batch), dtype=torch.uint8, device=device),
           "was_reset": torch.empty((steps, batch), dtype=torch.uint8, device=device)}
    for v in out.values():
        v.zero_()                                              # first touch of the fresh allocations, outside the timing
    stream = torch.cuda.current_stream(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    ev0.record(stream)
    env.rollout(acts, out, auto_reset=AUTO_RESET)
    ev1.record(stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    env.check_errors()
    n = batch * A * steps
    res = {"value": round(n / wall), "unit": "agent-steps/s", "steps": steps, "launches": 1,
           "ms_per_step": round(wall * 1e3 / steps, 6), "event_ms_per_step": round(ev0.elapsed_time(ev1) / steps, 6),
           "note": "mgx_rollout: K steps in one launch, bit-identical to K mgx_step calls (tests/test_hip_parity.py); "
                   "valid for open-loop action sequences such as this benchmark's random actions"}
    del env, out
    torch.cuda.empty_cache()
    return res


def cpu_baseline(spec, batch, budget_s=12.0):
    """Oracle (C port of the reference algorithm, OpenMP over envs) on this host, bounded sample."""
    from oracle import binding as ob
    cores = ob.max_threads()
    grid, agents = layouts.empty_layout(spec.width, spec.num_agents)
    from multigrid_amd import rng as rnglib
    st = dict(grid=np.repeat(grid[None], batch, 0).copy(), agents=np.repeat(agents[None], batch, 0).copy(),
              rng=rnglib.synthetic_words(batch, 1234), step_count=np.zeros(batch, np.int32))
    r = np.random.default_rng(1234)
    acts = r.integers(0, 7, size=(8, batch, spec.num_agents)).astype(np.int8)
    d = spec.as_dict()
    ob.step_batch(d, st["grid"], st["agents"], st["rng"], st["step_count"], acts[0], None, cores)   # warm
    n, t0 = 0, time.perf_counter()
    while True:
        ob.step_batch(d, st["grid"], st["agents"], st["rng"], st["step_count"], acts[n & 7], None, cores)
        n += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and n >= 8) or n >= 100000:
            break
    return {"value": round(n * batch * spec.num_agents / el), "unit": "agent-steps/s", "cores": cores,
            "kind": "port",
            "sample": f"{n} steps of the same workload (batch {batch}, Empty-16x16, 4 agents) = "
                      f"{n * batch * spec.num_agents} agent-steps in {el:.1f} s; oracle/mgx_oracle.c, OpenMP "
                      f"over envs, {cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4096, help="envs per GPU (BASELINE.json configs[1]: 4096)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: the K timed steps are one hipGraph replay; eager: K Python-level env.step calls")
    ap.add_argument("--settle-steps", type=int, default=3000,
                    help="untimed launches on a scratch env before the timed region (clock ramp); 0 disables")
    ap.add_argument("--no-auto-reset", action="store_true", help="step finished envs on as the reference does (base.py:408-409)")
    ap.add_argument("--large-batch", type=int, default=1 << 20)
    ap.add_argument("--no-extras", action="store_true", help="skip roofline_large / cpu_baseline legs")
    ap.add_argument("--skip-phases", type=int, default=0,
                    help="profiling probe: bit p set = the kernel skips phase Pp (results are then garbage)")
    ap.add_argument("--envs-per-wavefront", type=int, default=0,
                    help="tuning probe: override the launcher's choice of envs per wavefront (0 = automatic)")
    args = ap.parse_args()
    global AUTO_RESET
    AUTO_RESET = not args.no_auto_reset

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} "
                     f"--master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible); there is no CPU path to benchmark")
    # MGX_BENCH_ONE_GPU=1 (validation only): every rank on device 0 with gloo, to exercise the multi-rank code path on
    # a single-GPU box; the numbers of such a run mean nothing.
    one_gpu_check = os.environ.get("MGX_BENCH_ONE_GPU") == "1"
    device = torch.device("cuda", 0 if one_gpu_check else local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_check:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    def barrier():
        if dist is not None:
            if one_gpu_check:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    if args.skip_phases:
        from multigrid_amd import _lib
        _lib.lib().mgx_debug_skip_phases(args.skip_phases)
    if args.envs_per_wavefront:
        from multigrid_amd import _lib
        _lib.lib().mgx_debug_set_envs_per_wavefront(args.envs_per_wavefront)
    spec = workload_spec()
    B, A = args.batch, spec.num_agents
    first_env, count = shard_range(world * B, rank, world)      # weak scaling: every rank owns `--batch` envs
    assert count == B
    env = make_env(spec, B, device, first_env=first_env)
    warm = random_actions(max(args.warmup, 1), B, A, device, 1000 + rank)
    acts = random_actions(args.steps, B, A, device, 1234 + rank)
    for t in range(args.warmup):
        env.step(warm[t], auto_reset=AUTO_RESET)
    if args.settle_steps > 0:
        # The timed region is ~10 ms; the GPU needs longer than the W warm-up steps to reach its steady clocks.  More
        # untimed launches of the same kernel, on a scratch copy so that the measured envs' state is exactly "W steps in".
        scratch = make_env(spec, B, device, first_env=first_env)
        for t in range(args.settle_steps):
            scratch.step(warm[t % warm.shape[0]], auto_reset=AUTO_RESET)
        del scratch
    torch.cuda.synchronize(device)
    wall_s, ev_ms = timed_rollout(env, acts, args.mode, barrier)
    if not args.skip_phases:
        env.check_errors()

    t = torch.tensor([wall_s], dtype=torch.float64, device="cpu" if one_gpu_check else device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max = float(t.item())
    total_agent_steps = world * B * A * args.steps
    out = {
        "metric": "agent-steps/sec", "value": round(total_agent_steps / wall_max), "unit": "agent-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall_max * 1e3 / args.steps, 6), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"MultiGrid-Empty-16x16-v0 agents=4 view_size=7 batch={B} envs per GPU "
                               f"(BASELINE.json configs[1]), uniform random actions 0..6",
                   "batch_per_gpu": B, "global_batch": world * B, "agents": A, "grid": "16x16", "view_size": 7,
                   "mode": args.mode, "parallelism": f"env-sharded x{world}, no collective",
                   "launch": env.backend.launch_info(B),
                   "auto_reset": ("fused into the step launch (mgx_step_autoreset): envs that are done restart from the "
                                  "env class' fixed reset layout before the next step") if AUTO_RESET else False,
                   "clock_settle": f"{args.settle_steps} untimed steps on a scratch env before the timed region"},
    }
    if rank == 0:
        ms_launch = ev_ms / args.steps
        rf = roofline(B * A * spec.bytes_step(), ms_launch, pmc_traffic("step", B))
        rf.update(kernel="mgx_fused_kernel<7,step>", ms_per_launch=round(ms_launch, 5),
                  bytes_per_agent_step=spec.bytes_step(), algorithmic_bytes=B * A * spec.bytes_step(),
                  traffic_unit="bytes per launch (rocprofv3 PMC, profiles/traffic.json)",
                  note="working set fits the 256 MiB Infinity Cache at this batch: latency-bound, see roofline_large")
        out["roofline"] = rf
        if not args.no_extras and world == 1:                  # the large-batch / rollout / CPU legs: N=1 only
            r_step, r_obs = large_batch_points(spec, device, args.large_batch)
            out["roofline_large"] = r_step
            out["gen_obs_large"] = r_obs
            out["fused_rollout"] = rollout_point(spec, B, device, args.steps, first_env, 1234 + rank)
            out["aux_kernels"] = aux_kernel_points(device)
            out["cpu_baseline"] = cpu_baseline(spec, B)
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
