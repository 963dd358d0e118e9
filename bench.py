#!/usr/bin/env python3
"""bench.py -- throughput of the fused HIP step (action apply + gen_obs) on synthetic random-action rollouts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3|c5] [--mode graph|eager]

One "step" = one `MultiGridEnv.step` over the whole per-GPU batch = one launch of the fused kernel (auto-reset fused in).
Headline workload = BASELINE.json's north-star configuration C4: MultiGrid-Empty-16x16-v0, agents=4, view_size=7,
batch=65536 envs -- all of it on one GPU at N=1 (it is ~100 MB).  For N>1 (launched by `python -m torch.distributed.run
--nproc-per-node N ...`, one rank per GPU -- or plainly as `python bench.py --gpus N`, which then launches itself that way)
the configuration's batch is SPLIT over the ranks (`--scaling strong`, the default since round 5: BASELINE.json configs[3] read
literally -- 65536 envs sharded across the GPUs, 8192 per GPU at N=8, the shape the north star's 1e8 agent-steps/s target is stated
on; rank r owns the global envs [r x 65536/N, (r+1) x 65536/N) -- seeds and synthetic state are functions of the global env
index), and the SAME run then measures the weak point -- the configuration's batch on EVERY GPU, global batch N x 65536, flat by
construction -- and carries it in the line as `weak` (`--scaling weak` makes that the headline and carries the split as
`strong`).  The data path has NO collective -- envs never interact (SURVEY.md section
8e) -- and torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.

Timing.  W untimed warm-up steps, one untimed calibration replay, then the timed region: a hipGraph holding a whole
number of K-step blocks is replayed until the region is >= 50 ms (a K-step region alone would be ~0.5 ms at the
driver's K = 20); `ms_per_step` = region / steps in it, `timed_steps` says how many that was.  The region is bracketed by
barrier + torch.cuda.synchronize() on both sides, max over ranks.

`value` is the LOCK-STEP path: one launch of the fused kernel per step of the whole per-GPU batch, step t+1 enqueued behind step
t (what a policy that needs all B observations of step t before step t+1 gets; the reference's own semantics).  Beside it,
`pipelined`: the same steps issued as independent chains of sub-shard launches (BatchedMultiGridEnv.capture_steps(sub_shards=
"auto"), the product's own policy: mgx_sub_shards) -- valid for open-loop actions or a double-buffered actor loop.

Prints ONE JSON line on rank 0.  Extra objects (N=1 only, except `roofline`, `pipelined` and -- N>1 -- `weak` / `strong`):
  weak / strong       (N>1) the scaling point that is NOT the headline, measured in the same run: value, ms_per_step, per-rank
                      min / max, roofline of that launch (`weak` beside the default strong headline)
  roofline            the fused kernel on the timed workload, by the literal definition: algorithmic bytes per launch (SURVEY.md
                      8d) / average launch duration from HIP events over the timed region on the launch stream (one launch per
                      step, back to back; rocprofv3's average for the kernel: profiles/); `traffic` = HBM bytes per launch from the
                      committed rocprofv3 PMC passes (profiles/traffic.json)
  pipelined           the sub-sharded variant: value, ms_per_step, `step_frac` (the step's bytes / the step's time: an aggregate over
                      several launches in flight) and `launch` (ONE sub-shard launch over its own duration: the literal per-launch
                      fraction); the overlap itself is shown by in-kernel timestamps in profiles/r3_chain_overlap.txt
  configs             the other BASELINE.json GPU configurations (C2, C3 with its layout pool + hook, C5 with occluders),
                      each timed the same way, each with its own roofline
  eager               the same step called from Python once per step (policy-in-the-loop cost: ctypes + launch)
  cpu_baseline_python_1core  (round 5) SURVEY 8d(i): the pure Python / NumPy per-env restatement (oracle/py_oracle.py) on one host core
  fused_rollout       K steps as ONE mgx_rollout launch (open-loop actions only)
  roofline_large / gen_obs_large / one_hot_large / aux_kernels
                      the kernels at a working set >> the 256 MiB Infinity Cache (HBM-resident regime)
  cpu_baseline        the CPU oracle (a C port of the reference algorithm; the Python reference cannot travel to the GPU
                      box) on all host cores, bounded sample of the same workload;  cpu_baseline_1core: one thread
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from multigrid_amd import _lib, workloads  # noqa: E402
from multigrid_amd.sharding import shard_range  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
AUTO_RESET = True              # finished envs (all terminated / truncated) restart; fused into the step launch
MIN_REGION_MS = 50.0


def tool_workload() -> str:
    """Profiling tools pick their configuration with MGX_WORKLOAD=c2|c3|c4|c5 (default c2)."""
    return os.environ.get("MGX_WORKLOAD", "c2")


def tool_cell_bytes():
    """(tools/) MGX_CELL_BYTES=1|2 forces the grid's cell format; unset = the configuration's own (compact cells for C5)."""
    cb = os.environ.get("MGX_CELL_BYTES")
    return int(cb) if cb else None


def workload_spec():
    return workloads.spec_of(tool_workload(), tool_cell_bytes())


def make_env(spec, batch, device, first_env=0):
    """(tools/) `batch` envs of the MGX_WORKLOAD configuration starting at global env `first_env`."""
    name = tool_workload()
    wl = workloads.make(name, batch=batch, first_env=first_env,
                        global_batch=max(workloads.GLOBAL_BATCH[name], first_env + batch), cell_bytes=tool_cell_bytes())
    assert wl.spec == spec
    return wl.make_env(device, auto_reset=AUTO_RESET)


def random_actions(steps, batch, agents, device, seed):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 7, (steps, batch, agents), dtype=torch.int8, device=device, generator=g)


def capture_steps(env, actions, sub_shards=1):
    """One hipGraph holding `len(actions)` consecutive env.step launches (sub_shards=P: as P parallel chains over P
    consecutive blocks of the batch, BatchedMultiGridEnv.capture_steps)."""
    return env.capture_steps(actions, auto_reset=AUTO_RESET, sub_shards=sub_shards)


def timed_region(env, run_once, repeats, dist_barrier):
    """Time `repeats` calls of run_once (each enqueues a fixed number of steps).  Returns (wall s incl. sync, event ms)."""
    stream = torch.cuda.current_stream(env.device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dist_barrier()
    torch.cuda.synchronize(env.device)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(repeats):
        run_once()
    ev1.record(stream)
    torch.cuda.synchronize(env.device)
    t1 = time.perf_counter()
    dist_barrier()
    return t1 - t0, ev0.elapsed_time(ev1)


def measure_steps(env, K, warmup, mode, dist_barrier, seed, min_region_ms=MIN_REGION_MS, agree=None, sub_shards=1):
    """W warm-up steps, a calibration pass, then a timed region of whole K-step blocks lasting >= min_region_ms.
    Returns dict(wall_s, event_ms, timed_steps, blocks).  `agree(n)` lets all ranks settle on one repeat count."""
    B, A, dev = env.batch, env.spec.num_agents, env.device
    block = K * max(1, math.ceil(256 / K))                     # steps enqueued per run_once: >= 256, a multiple of K
    warm = random_actions(max(warmup, 1), B, A, dev, seed + 1000)
    for t in range(warmup):
        env.step(warm[t], auto_reset=AUTO_RESET)
    acts = random_actions(block, B, A, dev, seed)
    if mode == "graph":
        graph = capture_steps(env, acts, sub_shards)
        run_once = graph.replay
    else:
        def run_once():
            for t in range(block):
                env.step(acts[t], auto_reset=AUTO_RESET)
    _, cal_ms = timed_region(env, run_once, 2, dist_barrier)    # untimed for the result: clocks settle, gives the estimate
    repeats = max(1, math.ceil(1.25 * min_region_ms / max(cal_ms / 2, 1e-3)))      # (margin: the calibration pass runs slower)
    if agree is not None:
        repeats = agree(repeats)
    wall_s, ev_ms = timed_region(env, run_once, repeats, dist_barrier)
    return {"wall_s": wall_s, "event_ms": ev_ms, "timed_steps": block * repeats, "block": block, "repeats": repeats}


def kernel_time_ms(fn, iters, device, warm=30):
    """Average duration of `fn`'s single kernel launch: `iters` back-to-back launches between two HIP events, after
    `warm` untimed ones."""
    stream = torch.cuda.current_stream(device)
    for _ in range(warm):
        fn()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    ev0.record(stream)
    for _ in range(iters):
        fn()
    ev1.record(stream)
    torch.cuda.synchronize(device)
    return ev0.elapsed_time(ev1) / iters


def pmc_traffic(key: str, batch: int):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json: {key: {batch: {bytes}}}), or
    None if this (kernel, batch) was not profiled.  bench.py cannot collect PMC counters on itself."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            return json.load(fh)[key][str(batch)]["bytes"]
    except (OSError, KeyError, ValueError, TypeError):
        return None


def roofline(alg_bytes_per_launch, ms, traffic=None):
    achieved = alg_bytes_per_launch / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic}


def step_roofline(wl_name, spec, B, ms_launch):
    """Roofline of the fused step kernel, literal definition: one launch of B envs over its own duration."""
    alg = B * spec.num_agents * spec.bytes_step()
    # (the counter traffic of C5 on 16-bit cells -- the configuration's own format is the compact one -- has its own key)
    key = f"{wl_name}_wide" if (wl_name == "c5" and spec.cell_bytes == 2) else wl_name
    rf = roofline(alg, ms_launch, pmc_traffic(f"{key}_step", B))
    rf.update(kernel=f"mgx_fused_kernel<{spec.view_size},step,autoreset>", ms_per_launch=round(ms_launch, 5),
              bytes_per_agent_step=spec.bytes_step(), algorithmic_bytes=alg,
              traffic_unit="bytes per launch (rocprofv3 PMC, profiles/traffic.json)")
    return rf


def pipelined_point(wl_name, spec, env, K, device, barrier, seed, P, wall_scale=None, min_region_ms=30.0, agree=None, G=None):
    """The same steps as P independent chains of sub-shard launches (capture_steps(sub_shards=P))."""
    B, A = env.batch, spec.num_agents
    # (graphs of at most ~256 steps per chain, replayed: a replay's host side has to feed P queues, and on graphs of thousands of
    # kernel nodes it no longer keeps up -- profiles/r5_chain_policy.txt)
    Kp = K if K <= 256 else 250
    m = measure_steps(env, Kp, 0, "graph", barrier, seed=seed, min_region_ms=min_region_ms, sub_shards=P, agree=agree)
    ms_step = m["event_ms"] / m["timed_steps"]
    ms_launch = sub_shard_launch_ms(env, P, device)
    alg = B * A * spec.bytes_step()
    part = pmc_traffic(f"{wl_name}_part_step", B // P)
    one = roofline(alg // P, ms_launch, part)
    return m, {"sub_shards": P, "ms_per_step": round(ms_step, 6), "timed_steps": m["timed_steps"],
               "step_frac": round(alg / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "step_achieved_GBs": round(alg / (ms_step * 1e-3) / 1e9, 1),
               "launch": {"batch": B // P, "algorithmic_bytes": alg // P, "ms_per_launch": round(ms_launch, 5),
                          "achieved": one["achieved"], "frac": one["frac"], "traffic": part},
               "mean_launches_in_flight": round(P * ms_launch / ms_step, 2),
               "note": f"a step of the batch = {P} launches of {B // P} envs from {P} independent chains on {P} streams (parallel "
                       "branches of one hipGraph), several in flight at a time; `step_frac` = the step's algorithmic bytes / the "
                       "step's time (an aggregate: launch.frac x launches in flight, computed), `launch` = one such launch over its "
                       "own duration, alone on the chip; the overlap OBSERVED with in-kernel timestamps: profiles/r3_chain_overlap.txt "
                       "(rocprofv3 serialises the chains).  Open-loop actions or a double-buffered actor loop only."}


def sub_shard_launch_ms(env, P, device):
    """Duration of one sub-shard launch: HIP events over back-to-back launches of the first of `P` sub-shards, alone."""
    sh = env.split(P)[0]
    acts = random_actions(4, sh.batch, env.spec.num_agents, device, 5)
    i = [0]

    def step():
        sh.step(acts[i[0] & 3], auto_reset=AUTO_RESET); i[0] += 1
    return kernel_time_ms(step, 200, device, warm=50)


def config_point(name, device, K, warmup, device_generated=False, steady=False, cell_bytes=None):
    """One of the other BASELINE.json configurations, all of it on this GPU, timed like the headline (lock-step: one launch per
    step); `pipelined` beside it when the product's policy suggests sub-shards for it.
    device_generated: episode starts are generated ON THE DEVICE (the reference's _gen_grid with numpy-exact draws, in the
    tail of the step's own launch: mgx_step_generate) instead of picked from the host-made layout pool."""
    wl = workloads.make(name, cell_bytes=cell_bytes)
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    if device_generated:
        # (set_layout_generator's default: every episode end adopts a candidate made while its episode ran -- one per value of the
        # generator's np_random draw -- by generator launches between the steps, one every max_steps / 3 steps)
        # steady: the episodes are OUT OF PHASE (uniform over the episode length: what any long rollout settles into -- every step
        # then sees its share of truncations, ~B / max_steps, instead of one burst every max_steps steps)
        env.set_layout_generator("blockedunlockpickup", layout_seed=5, room_size=6)
        if steady:
            env.step_count.copy_(torch.arange(wl.batch, device=device, dtype=torch.int32) % wl.spec.max_steps)
    m = measure_steps(env, K, warmup, "graph", lambda: None, seed=4321, min_region_ms=30.0)
    env.check_errors()
    B, A = wl.batch, wl.spec.num_agents
    ms = m["event_ms"] / m["timed_steps"]
    out = {"workload": wl.title, "batch": B, "agents": A, "grid": f"{wl.spec.width}x{wl.spec.height}",
           "view_size": wl.spec.view_size, "cell_bytes": wl.spec.cell_bytes,
           "ms_per_step": round(m["wall_s"] * 1e3 / m["timed_steps"], 6),
           "value": round(B * A * m["timed_steps"] / m["wall_s"]), "unit": "agent-steps/s",
           "timed_steps": m["timed_steps"],
           "layout_pool": ("generated on the device in the step's own launch (mgx_step_generate)"
                           + "; every episode end adopts one of 4 candidates (one per door row) made while the episode ran, by a "
                             "generator launch between two steps every max_steps / 3 steps (MgxGenStage.candidates)"
                           + ("; episodes out of phase" if steady else ""))
                          if device_generated else int(wl.pool[0].shape[0]),
           "resets_in_region": int(env.episode.sum().item()) if AUTO_RESET else 0,
           "launch": env.backend.launch_info(B),
           "roofline": step_roofline(name, wl.spec, B, ms)}
    P = env.sub_shards_hint(AUTO_RESET)
    if P > 1:
        m2, pp = pipelined_point(name, wl.spec, env, K, device, lambda: None, 4322, P, min_region_ms=20.0)
        pp["value"] = round(B * A * m2["timed_steps"] / m2["wall_s"])
        out["pipelined"] = pp
    del env
    torch.cuda.empty_cache()
    return out


def short_episode_point(device, max_steps=32, lead=32):
    """C3's shape with episodes of at most `max_steps` steps, out of phase (B / max_steps envs restart EVERY step: 512 at 32), each
    restart generated on the device.  staged="candidates" with a generator launch every lead / 2 steps (round 6: that launch is
    ~15 us, mgx_layout_gen.h place_group) against generation in the tail of the step's own launch.  `adopted`: of the restarts of an
    eager, untimed pass, the share whose candidates were all current when the episode ended (those copy a staged layout; the others
    run the generator in the step's tail) -- candidates do not depend on WHEN the episode ends, so an early end adopts like a
    truncation as long as a generator launch came by since the env's previous restart."""
    import dataclasses
    from multigrid_amd.batched import BatchedMultiGridEnv
    wl = workloads.make("c3")
    spec = dataclasses.replace(wl.spec, max_steps=max_steps)
    B, A = wl.batch, spec.num_agents
    out = {"workload": f"{wl.title}, max_steps={max_steps}", "batch": B, "restarts_per_step": B // max_steps}
    for key, staged in (("candidates", "candidates"), ("in_tail", False)):
        env = BatchedMultiGridEnv(spec, B, device)
        env.load_state(wl.grid, wl.agents, rng=wl.rng, aux=wl.aux, validate=False)
        env.set_layout_generator("blockedunlockpickup", layout_seed=5, room_size=6, staged=staged, lead=lead if staged else None)
        env.step_count.copy_(torch.arange(B, device=device, dtype=torch.int32) % max_steps)
        if staged:                                              # the untimed accounting pass (also the warm-up)
            acts = random_actions(4 * max_steps, B, A, device, 77)
            ready_n = total_n = 0
            for t in range(4 * max_steps):
                tag = env._gen["stage"]["tag"]
                ready = (tag == env.episode[:, None]).all(1)
                env.step(acts[t], auto_reset=True)
                if t >= max_steps:                              # (the first episode's candidates come with the first generator launch)
                    wr = env.was_reset.bool()
                    ready_n += int((ready & wr).sum()); total_n += int(wr.sum())
            out["adopted"] = round(ready_n / max(total_n, 1), 4)
            out["restarts_counted"] = total_n
        m = measure_steps(env, 256, 50, "graph", lambda: None, seed=4321, min_region_ms=30.0)
        env.check_errors()
        out[key + "_us_per_step"] = round(m["event_ms"] / m["timed_steps"] * 1e3, 3)
        del env
        torch.cuda.empty_cache()
    out["generator_launch_every_steps"] = lead // 2
    return out


def one_hot_config_point(name, device, cell_bytes=None, T=128):
    """A configuration stepped with ONE-HOT observations (u8[B,A,v,v,21], written by the step's own launch) -- what RLlib's default
    registration of the reference feeds the policy (multigrid/rllib/__init__.py:110-111) -- as hipGraph replays, fused auto-reset.
    `cell_bytes`: the grid format (round 6: compact cells serve this step too)."""
    wl = workloads.make(name, cell_bytes=cell_bytes)
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    B, A = wl.batch, wl.spec.num_agents
    acts = random_actions(T, B, A, device, 4321)
    for t in range(30):
        env.step(acts[t], auto_reset=AUTO_RESET, one_hot=True)
    g = env.capture_steps(acts, auto_reset=AUTO_RESET, one_hot=True)
    g.replay()
    ms = kernel_time_ms(g.replay, 3, device, warm=1) / T
    env.check_errors()
    v2 = wl.spec.view_size ** 2
    algo = B * A * (wl.spec.bytes_step() + 18 * v2)            # the step's bytes with a 21-byte instead of a 3-byte observation cell
    out = {"workload": wl.title, "batch": B, "cell_bytes": wl.spec.cell_bytes, "ms_per_step": round(ms, 6),
           "value": round(B * A / (ms * 1e-3)), "unit": "agent-steps/s", "timed_steps": 3 * T,
           "algorithmic_bytes": algo, "frac": round(algo / (ms * 1e-3) / 8e12, 4), "launch": env.backend.launch_info(B)}
    del env, g
    torch.cuda.empty_cache()
    return out


def eager_point(wl, device, steps=2000, sub_shards=1):
    """env.step called from Python once per step (what an RL loop that cannot capture its policy pays).
    sub_shards="auto": the eager sub-shard form (mgx_step_chains on the env's side streams, joined once at the end)."""
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    B, A = wl.batch, wl.spec.num_agents
    acts = list(random_actions(64, B, A, device, 77))           # (64 action tensors: a policy hands over a tensor, not a slice)
    for t in range(200):
        env.step(acts[t & 63], auto_reset=AUTO_RESET, sub_shards=sub_shards)
    env.join()
    stream = torch.cuda.current_stream(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    ev0.record(stream)
    for t in range(steps):
        env.step(acts[t & 63], auto_reset=AUTO_RESET, sub_shards=sub_shards)
    t_host = time.perf_counter() - t0
    env.join()
    ev1.record(stream)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    env.check_errors()
    out = {"workload": wl.name, "batch": B, "steps": steps, "ms_per_step": round(wall * 1e3 / steps, 6),
           "host_ms_per_call": round(t_host * 1e3 / steps, 6), "event_ms_per_step": round(ev0.elapsed_time(ev1) / steps, 6),
           "value": round(B * A * steps / wall), "unit": "agent-steps/s",
           "sub_shards": env.sub_shards_hint(AUTO_RESET, form="eager") if sub_shards == "auto" else sub_shards,
           "note": "BatchedMultiGridEnv.step from Python per step (one ctypes call: mgx_step_ex"
                   + (" / mgx_step_chains, the chains joined once after the last step" if sub_shards != 1 else "") + "); no graph"}
    del env
    return out


def large_batch_points(device, large_batch):
    wl = workloads.make("c4", batch=large_batch, global_batch=large_batch)
    spec = wl.spec
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    acts = random_actions(4, large_batch, spec.num_agents, device, 99)
    i = [0]

    def step():
        env.step(acts[i[0] & 3], auto_reset=AUTO_RESET); i[0] += 1
    # 250 untimed steps first: right after the start all agents of an env stand on one cell, which sends most envs
    # through the sequential fallback (343 us per step over the first 50 steps, 270 us from step 150 on: tools/time_curve.py);
    # the timed steps are the steady state of the random-action rollout, as the headline's are
    ms_step = kernel_time_ms(step, 60, device, warm=250)
    ms_obs = kernel_time_ms(env.gen_obs, 60, device)
    n = large_batch * spec.num_agents
    r_step = roofline(n * spec.bytes_step(), ms_step, pmc_traffic("large_step", large_batch))
    r_step.update(batch=large_batch, kernel="mgx_fused_kernel<7,step,autoreset>", ms_per_launch=round(ms_step, 4),
                  agent_steps_per_s=round(n / (ms_step * 1e-3)))
    r_obs = roofline(n * spec.bytes_gen_obs(), ms_obs, pmc_traffic("large_gen_obs", large_batch))
    r_obs.update(batch=large_batch, kernel="mgx_fused_kernel<7,gen_obs>", ms_per_launch=round(ms_obs, 4),
                 agent_views_per_s=round(n / (ms_obs * 1e-3)))
    out = {"roofline_large": r_step, "gen_obs_large": r_obs}
    # the step with the observation written one-hot encoded by the same launch (mgx_step_one_hot; what RLlib's default
    # OneHotObsWrapper registration consumes): algorithmic bytes = bytes_step with the 3 v^2 obs bytes replaced by 21 v^2
    def step_oh():
        env.step(acts[i[0] & 3], auto_reset=AUTO_RESET, one_hot=True); i[0] += 1
    ms_oh = kernel_time_ms(step_oh, 30, device, warm=10)
    by_oh = spec.bytes_step() + 18 * spec.view_size ** 2
    r_oh = roofline(n * by_oh, ms_oh, pmc_traffic("large_step_one_hot", large_batch))
    r_oh.update(batch=large_batch, kernel="mgx_fused_kernel<7,step,autoreset,one_hot>", ms_per_launch=round(ms_oh, 4),
                bytes_per_agent_step=by_oh, agent_steps_per_s=round(n / (ms_oh * 1e-3)))
    out["one_hot_large"] = r_oh
    out["aux_kernels"] = aux_kernel_points(env, device)
    out["one_hot_large"]["unfused_ms"] = round(ms_step + out["aux_kernels"]["one_hot"]["ms_per_launch"], 4)
    del env
    torch.cuda.empty_cache()
    return out


def aux_kernel_points(env, device):
    """Bandwidth of the kernels either side of the fused step at an HBM-resident size."""
    spec, batch = env.spec, env.batch
    A, V, H, W = spec.num_agents, spec.view_size, spec.height, spec.width
    out = {}

    def entry(t, by):
        return {"batch": batch, "ms_per_launch": round(t, 5), "algorithmic_bytes": by,
                "achieved_GBs": round(by / t / 1e6, 1), "frac": round(by / t / 1e6 / HBM_PEAK_GBS, 4)}
    # one-hot of the step's observations: 3 B in, 21 B out per view cell (wrappers.py:158-190)
    cells = batch * A * V * V
    out["one_hot"] = entry(kernel_time_ms(env.one_hot_obs, 30, device, warm=10), cells * (3 + 21))
    # fully observable encode: packed grid in (2 B per cell), transposed (type, color, state) bytes out, agent rows in
    out["full_obs"] = entry(kernel_time_ms(env.full_obs, 30, device, warm=10), batch * (H * W * (2 + 3) + A * 8))
    # auto-reset with every env done: agent rows + step counts in, layout out
    def reset_all():
        env.step_count.fill_(spec.max_steps)          # every env truncated -> every env is reset
        env.reset_done()
    def fill_only():
        env.step_count.fill_(spec.max_steps)
    t = kernel_time_ms(reset_all, 30, device, warm=10) - kernel_time_ms(fill_only, 30, device, warm=10)
    out["reset_done_all"] = entry(t, batch * (H * W * 2 + 2 * A * 8 + 4 + 4 + 4 + 1))
    env.step_count.zero_()
    out["reset_done_none"] = entry(kernel_time_ms(env.reset_done, 30, device, warm=10), batch * (A * 8 + 4 + 1))
    return out


def generation_point(device, batch=1 << 18):
    """mgx_reset_generate on BlockedUnlockPickup envs: the scan alone (nobody done) and with EVERY env restarting (each runs
    the reference's _gen_grid: ~25 numpy-compatible draws + rejection sampling per episode)."""
    wl = workloads.make("c3", batch=batch, global_batch=batch)
    env = wl.make_env(device, auto_reset=False)
    env.set_layout_generator("blockedunlockpickup", layout_seed=5, room_size=6)
    spec = wl.spec
    t_none = kernel_time_ms(env.reset_done, 30, device, warm=10)

    def reset_all():
        env.step_count.fill_(spec.max_steps)
        env.reset_done()
    def fill_only():
        env.step_count.fill_(spec.max_steps)
    t_all = kernel_time_ms(reset_all, 20, device, warm=5) - kernel_time_ms(fill_only, 20, device, warm=5)
    assert int(env.episode.min().item()) >= 25
    del env
    torch.cuda.empty_cache()
    return {"batch": batch, "scan_ms": round(t_none, 5), "all_envs_regenerated_ms": round(t_all, 5),
            "episodes_per_s": round(batch / (t_all * 1e-3)),
            "note": "BlockedUnlockPickup 11x6, 2 agents: one lane per env runs the reference's _gen_grid on the device"}


def rollout_point(wl, device, steps, obs_budget=1 << 30):
    """`steps` steps as ONE mgx_rollout launch (env state stays in LDS between steps).  Open-loop actions only."""
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    spec, batch = wl.spec, wl.batch
    steps = max(2, min(steps, obs_budget // max(1, batch * spec.num_agents * spec.view_size ** 2 * 3)))   # obs[T] <= 1 GiB by default
    for t, a in enumerate(random_actions(60, batch, spec.num_agents, device, 99)):      # (leave the cold start: every agent on one cell)
        env.step(a, auto_reset=AUTO_RESET)
    acts = random_actions(steps, batch, spec.num_agents, device, 1234)
    out = env.rollout(acts[:2].contiguous(), auto_reset=AUTO_RESET)   # warm-up + allocation pattern
    A, v = spec.num_agents, spec.view_size
    out = {"obs": torch.empty((steps, batch, A, v, v, 3), dtype=torch.uint8, device=device),
           "dir": torch.empty((steps, batch, A), dtype=torch.uint8, device=device),
           "reward": torch.empty((steps, batch, A), dtype=torch.float64, device=device),
           "terminated": torch.empty((steps, batch, A), dtype=torch.uint8, device=device),
           "truncated": torch.empty((steps, batch), dtype=torch.uint8, device=device),
           "was_reset": torch.empty((steps, batch), dtype=torch.uint8, device=device)}
    for x in out.values():
        x.zero_()                                              # first touch of the fresh allocations, outside the timing
    stream = torch.cuda.current_stream(device)
    wall, ev_ms = float("inf"), float("inf")
    for rep in range(2):                                       # (the second launch: every page of obs[T] touched by a kernel before)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        ev0.record(stream)
        env.rollout(acts, out, auto_reset=AUTO_RESET)
        ev1.record(stream)
        torch.cuda.synchronize(device)
        wall = min(wall, time.perf_counter() - t0)
        ev_ms = min(ev_ms, ev0.elapsed_time(ev1))
    env.check_errors()
    n = batch * A * steps
    from multigrid_amd import _lib
    res = {"workload": wl.name, "batch": batch, "value": round(n / wall), "unit": "agent-steps/s", "steps": steps,
           "launches": 1, "ms_per_step": round(wall * 1e3 / steps, 6),
           "event_ms_per_step": round(ev_ms / steps, 6),
           "launch": _lib.launch_info(spec, batch, roll=True),
           "frac_of_hbm_peak_at_the_step_kernels_bytes": round(batch * spec.num_agents * spec.bytes_step() / (ev_ms / steps * 1e-3) / 8e12, 4),
           "note": "mgx_rollout: the steps in one launch, bit-identical to that many mgx_step calls "
                   "(tests/test_hip_parity.py); valid for open-loop action sequences such as this benchmark's random actions"}
    del env, out
    torch.cuda.empty_cache()
    return res


def persistent_point(name, base, batch, device, T=400):
    """Closed-loop stepping with ONE resident launch (mgx_step_persistent) fed by the shortest producer there can be
    (mgx_persistent_feed: wait for step t-1's flags, post step t's granules), from the producer's own s_memrealtime trace; beside it
    the same steps as one launch per step with ONE copy kernel per step as the policy (hipGraph).  In-kernel spans per wavefront
    (granules seen -> flag stored): profiles/r4_persistent.txt."""
    wl = workloads.make(base, batch=batch, global_batch=max(batch, workloads.GLOBAL_BATCH[base]))
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    B, A = wl.batch, wl.spec.num_agents
    acts = random_actions(T, B, A, device, 1234)
    for t in range(50):
        env.step(acts[t], auto_reset=AUTO_RESET)
    for _ in range(2):
        with env.persistent(max_steps=T, auto_reset=AUTO_RESET) as ps:
            tr = ps.feed(acts, trace=True)
    torch.cuda.synchronize(device)
    tr = tr.cpu().numpy().astype(np.int64) / 100.0                 # us
    seen, posted = tr[0::2], tr[1::2]
    k = T // 5
    step_us = float(np.diff(posted)[k:].mean())
    # the launches' side of the A/B: [copy kernel -> step launch] x T in one graph
    cur = torch.zeros((B, A), dtype=torch.int8, device=device)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device)
    s.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for t in range(T):
                cur.copy_(acts[t])
                env.step(cur, auto_reset=AUTO_RESET)
    torch.cuda.current_stream(device).wait_stream(s)
    g.replay()
    ms = kernel_time_ms(g.replay, 5, device, warm=1)
    out = {"workload": wl.title, "batch": B, "wavefronts_resident": ps.waves,
           "us_per_step": round(step_us, 3), "value": round(B * A / (step_us * 1e-6)), "unit": "agent-steps/s",
           "env_us": round(float((seen[1:] - posted)[k:].mean()), 3), "producer_us": round(float((posted - seen[:-1])[k:].mean()), 3),
           "launches_with_a_policy_kernel_us_per_step": round(ms * 1e3 / T, 3),
           "note": "closed loop: env_us = granules of step t posted -> all flags of step t seen by the producer (the env's chain + "
                   "write-through drain + the producer's flag sweep), producer_us = flags seen -> next granules posted; against one launch "
                   "per step with one copy kernel per step as policy (two launch boundaries per step); `value`'s open-loop graph number "
                   "has no policy and no hand-off"}
    del env
    torch.cuda.empty_cache()
    return out


def byte_grid_overhead(wl, device):
    """The step through torch.ops.mgx.step_out on the reference's grid form, u8[B,H,W,3] -- since round 5 handed to the kernel as it is
    (MgxSpec.cell_bytes = 3: packed into the LDS tile by the step's own P0, changed cells written back as bytes) --, against the same
    op on packed cells."""
    from multigrid_amd import ops
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    ints = ops.spec_to_ints(wl.spec)
    acts = random_actions(4, wl.batch, wl.spec.num_agents, device, 5)
    grid_u8 = env.grid.contiguous()
    i = [0]

    def packed():
        torch.ops.mgx.step_out(env.cells, env.agents, env.rng, env.step_count, acts[i[0] & 3], None, env.err, ints, env.obs, env.dir,
                               env.reward, env.terminated, env.truncated); i[0] += 1

    def as_bytes():
        torch.ops.mgx.step_out(grid_u8, env.agents, env.rng, env.step_count, acts[i[0] & 3], None, env.err, ints, env.obs, env.dir,
                               env.reward, env.terminated, env.truncated); i[0] += 1
    t_p = kernel_time_ms(packed, 200, device, warm=50)
    t_b = kernel_time_ms(as_bytes, 200, device, warm=50)
    torch.ops.mgx.check_errors(device.index or 0)
    del env
    torch.cuda.empty_cache()
    return {"workload": wl.name, "batch": wl.batch, "packed_cells_ms_per_step": round(t_p, 5), "byte_grid_ms_per_step": round(t_b, 5),
            "overhead_ms": round(t_b - t_p, 5),
            "note": "torch.ops.mgx.step_out eager, no auto-reset; byte grid u8[B,H,W,3] = the same launch with the conversion (and the "
                    "wall-ring check) in its load phase (round 4: a pack and an unpack launch around the step, +36..43 us)"}


def cpu_baseline_python(wl, budget_s=4.0, sample_envs=4):
    """the pure Python / NumPy per-env restatement (oracle/py_oracle.py, pinned to the reference's fixtures) on ONE
    host core, a few envs of the timed workload -- the stand-in for the reference's own speed, whose two hot kernels are numba
    functions that run as exactly such interpreter loops without numba (BASELINE.md section 2 measured the real reference at
    ~5.4e3 agent-steps/s that way)."""
    from multigrid_amd import layouts
    from oracle import py_oracle as po
    n = min(sample_envs, wl.batch)
    spec, A = wl.spec.as_dict(), wl.spec.num_agents
    envs = [(layouts.grid_from_product(wl.grid[b]), layouts.unpack_agents(wl.agents[b]), np.random.default_rng(b), 0) for b in range(n)]
    r = np.random.default_rng(1)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for k, (g, a, rng, sc) in enumerate(envs):
            out = po.step(spec, g, a, rng, sc, r.integers(0, 7, size=A), None if wl.aux is None else wl.aux[k])
            envs[k] = (g, a, rng, out[5])
        steps += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * A * steps / dt), "unit": "agent-steps/s", "cores": 1, "kind": "port",
            "sample": f"{steps} steps of the first {n} envs of the timed workload ({wl.name}) = {n * A * steps} agent-steps in {dt:.1f} s; "
                      "oracle/py_oracle.py: the reference's algorithm restated per env in Python / NumPy (interpreter loops for gen_obs "
                      "and the visibility sweep, as the reference's numba kernels run without numba), no auto-reset"}


def c1_point(device, T=256):
    """BASELINE.json configs[0]: MultiGrid-Empty-8x8-v0, 2 agents, ONE env, through the reference's own surface -- `MultiGridEnv.reset`
    / `step(dict) -> 5 dicts` (multigrid_amd/env.py over the HIP kernels) -- with the survey's C1 inputs (seed 0, actions
    default_rng(0).integers(0, 7, (T, A))).  What a drop-in user of the dict API pays per step, split into its parts:
        launch_us     actions host -> device + the fused kernel's launch + its completion (BatchedMultiGridEnv.step, synchronised)
        d2h_us        ONE device-to-host copy of reward | obs | dir | terminated | truncated | err (outputs_to_host, pinned buffer)
        dicts_us      the rest of env.step: building the five dicts, int64 images
    beside oracle/py_oracle.py (the reference's algorithm per env in Python / NumPy, the stand-in for the reference's own speed
    without numba) on the same episode on one host core."""
    import multigrid_amd as mg
    from multigrid_amd import layouts
    from oracle import py_oracle as po
    A = 2
    acts = np.random.default_rng(0).integers(0, 7, (T, A))
    env = mg.make("MultiGrid-Empty-8x8-v0", agents=A, device=str(device))
    env.reset(seed=0)
    for t in range(20):                                        # warm-up (the first launches, the pinned buffer)
        env.step({i: int(acts[t, i]) for i in range(A)})
    env.reset(seed=0)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for t in range(T):
        env.step({i: int(acts[t, i]) for i in range(A)})
    total = (time.perf_counter() - t0) / T
    # the parts, each timed by itself over the same episode
    benv = env._benv
    env.reset(seed=0)
    dev_acts = [torch.from_numpy(acts[t].astype(np.int8)[None]) for t in range(T)]
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for t in range(T):
        benv.step(dev_acts[t].to(benv.device))
        torch.cuda.current_stream(device).synchronize()
    launch = (time.perf_counter() - t0) / T
    t0 = time.perf_counter()
    for t in range(T):
        benv.outputs_to_host()
    d2h = (time.perf_counter() - t0) / T
    # the Python restatement on the same episode, one core
    spec = env.spec.as_dict()
    env.reset(seed=0)
    g = layouts.grid_from_product(benv.grid[0].cpu().numpy())
    a9 = layouts.unpack_agents(benv.agents[0].cpu().numpy())
    rng, sc = np.random.Generator(np.random.PCG64(np.random.SeedSequence(0))), 0
    t0 = time.perf_counter()
    for t in range(T):
        out = po.step(spec, g, a9, rng, sc, acts[t], None)
        sc = out[5]
    py = (time.perf_counter() - t0) / T
    del env
    return {"workload": workloads.TITLES["c1"], "steps": T, "us_per_step": round(total * 1e6, 2),
            "value": round(A / total), "unit": "agent-steps/s",
            "launch_us": round(launch * 1e6, 2), "d2h_us": round(d2h * 1e6, 2),
            "dicts_us": round(max(0.0, total - launch - d2h) * 1e6, 2), "d2h_copies_per_step": 1,
            "python_restatement_us_per_step": round(py * 1e6, 2), "python_restatement_value": round(A / py),
            "note": "the reference's dict API on ONE env is host-bound by construction (a launch and a copy per step for 2 agents); "
                    "the batched engine is the throughput path -- this is what the drop-in surface costs, not a roofline point"}


def cpu_baseline(wl, threads, budget_s, sample_envs):
    """Oracle (C port of the reference algorithm, OpenMP over envs) on this host, bounded sample of the workload."""
    from oracle import binding as ob
    n_env = min(sample_envs, wl.batch)
    st = dict(grid=wl.grid[:n_env].copy(), agents=wl.agents[:n_env].copy(), rng=wl.rng[:n_env].copy(),
              step_count=np.zeros(n_env, np.int32), aux=None if wl.aux is None else wl.aux[:n_env].copy())
    A = wl.spec.num_agents
    acts = np.random.default_rng(1234).integers(0, 7, size=(8, n_env, A)).astype(np.int8)
    d = wl.spec.as_dict()
    out = ob.step_outputs(d, n_env)                             # reused: the timing is the oracle, not page faults
    ob.step_batch(d, st["grid"], st["agents"], st["rng"], st["step_count"], acts[0], st["aux"], threads, out)   # warm
    n, t0 = 0, time.perf_counter()
    while True:
        ob.step_batch(d, st["grid"], st["agents"], st["rng"], st["step_count"], acts[n & 7], st["aux"], threads, out)
        n += 1
        el = time.perf_counter() - t0
        if (el >= budget_s and n >= 4) or n >= 100000:
            break
    return {"value": round(n * n_env * A / el), "unit": "agent-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} steps of the first {n_env} envs of the timed workload ({wl.name}: {wl.title}) = "
                      f"{n * n_env * A} agent-steps in {el:.1f} s; oracle/mgx_oracle.c, OpenMP over envs, {threads} thread(s), "
                      f"no auto-reset (as the reference).  A C port of the reference's algorithm, NOT a proxy for the reference's "
                      f"speed: the reference itself (Python, interpreter mode, 1 core) measured ~5.4e3 agent-steps/s in the build "
                      f"container (BASELINE.md), ~580x below this port on one core"}


def self_launch(n_gpus: int) -> int:
    """Re-run this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`
    (what the driver's multi-GPU form is) and hand its output through; returns the launcher's exit code."""
    import socket
    import subprocess
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (no HIP device visible); there is no CPU path to benchmark", file=sys.stderr)
        return 1
    have = torch.cuda.device_count()
    if have < n_gpus and os.environ.get("MGX_BENCH_ONE_GPU") != "1":
        print(f"--gpus {n_gpus}: only {have} HIP device(s) visible on this node (MGX_BENCH_ONE_GPU=1 runs every rank on "
              f"device 0 to validate the plumbing; its numbers mean nothing)", file=sys.stderr)
        return 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # (dmabuf IPC: what RCCL needs on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c4",
                    help="BASELINE.json configuration timed as `value` (default c4, the north-star configuration)")
    ap.add_argument("--global-batch", type=int, default=0, help="override the configuration's batch (all GPUs together)")
    ap.add_argument("--mode", choices=["graph", "eager"], default="graph",
                    help="graph: the timed steps are hipGraph replays; eager: Python-level env.step calls")
    ap.add_argument("--no-auto-reset", action="store_true", help="step finished envs on as the reference does (base.py:408-409)")
    ap.add_argument("--sub-shards", type=int, default=1,
                    help="graph mode: the chains the HEADLINE steps the batch as (default 1: lock-step, one launch per step; 0 = "
                         "the product's policy, BatchedMultiGridEnv.sub_shards_hint).  The pipelined variant is reported "
                         "beside the headline either way")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong (default) = the configuration's batch split over the N GPUs (BASELINE.json configs[3] read "
                         "literally: 65 536 envs over 8 GPUs, the shape the 1e8 agent-steps/s target is stated on); weak = every "
                         "GPU steps the configuration's batch (per-GPU work fixed, the global batch grows with N; envs are "
                         "independent, no collective).  The run also measures the OTHER point and carries it in the line as "
                         "`weak` / `strong`")
    ap.add_argument("--large-batch", type=int, default=1 << 20)
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement + its roofline")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the sub-sharded variant beside the headline (profiling: its launches are instances of the same kernel "
                         "at half the batch and would mix into the kernel's rocprofv3 statistics)")
    args = ap.parse_args()
    global AUTO_RESET
    AUTO_RESET = not args.no_auto_reset

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, rendezvous on 127.0.0.1) -- the
        # same command line the driver's `python -m torch.distributed.run ... bench.py --gpus N` form runs; rank 0 prints the line
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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

    def all_max(x: float) -> float:
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if one_gpu_check else device)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    name = args.workload
    G0 = args.global_batch or workloads.GLOBAL_BATCH[name]       # the configuration's batch
    G = G0 * world if args.scaling == "weak" else G0             # weak: that batch on EVERY GPU; strong: split over the GPUs
    first_env, B = shard_range(G, rank, world)                   # (seeds / synthetic state are functions of the GLOBAL env index)
    wl = workloads.make(name, batch=B, first_env=first_env, global_batch=G)
    spec, A = wl.spec, wl.spec.num_agents
    env = wl.make_env(device, auto_reset=AUTO_RESET)
    hint = int(all_max(float(env.sub_shards_hint(AUTO_RESET))))          # the product's policy (mgx_sub_shards), same on all ranks
    P = 1 if args.mode != "graph" else (args.sub_shards or hint)
    m = measure_steps(env, args.steps, args.warmup, args.mode, barrier, seed=1234 + rank,
                      agree=lambda n: int(all_max(float(n))), sub_shards=P)
    env.check_errors()
    wall_max = all_max(m["wall_s"])
    wall_min = -all_max(-m["wall_s"])
    S = m["timed_steps"]
    valid = _lib.is_product_lib()
    out = {
        "metric": "agent-steps/sec", "value": round(G * A * S / wall_max), "unit": "agent-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall_max * 1e3 / S, 6), "higher_is_better": True, "scaling": args.scaling,
        "ms_per_step_ranks": {"min": round(wall_min * 1e3 / S, 6), "max": round(wall_max * 1e3 / S, 6)},
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "timed_steps": S, "timed_region_ms": round(wall_max * 1e3, 3),
        "config": {"workload": wl.title + ", uniform random actions 0..6", "name": name, "cell_bytes": spec.cell_bytes,
                   "global_batch": G, "batch_per_gpu": B, "agents": A, "grid": f"{spec.width}x{spec.height}",
                   "view_size": spec.view_size, "mode": args.mode,
                   "configuration_batch": G0,
                   "parallelism": (f"env-sharded x{world}, no collective: " +
                                   (f"weak scaling, the configuration's {G0} envs on every GPU" if args.scaling == "weak" else
                                    f"strong scaling, the configuration's {G0} envs split over the GPUs")),
                   "sub_shards": P,
                   "semantics": ("lock-step: one launch of the fused kernel per step of the whole per-GPU batch, step t+1 behind "
                                 "step t" if P == 1 else
                                 "pipelined: independent chains of sub-shard launches (see `pipelined.note`)"),
                   "launch": env.backend.launch_info(B // P),
                   "auto_reset": ("fused into the step launch (mgx_step_autoreset): envs that are done restart from the "
                                  "layout pool before the next step") if AUTO_RESET else False,
                   "layout_pool": int(wl.pool[0].shape[0]),
                   "timing": f"{m['repeats']} x {m['block']}-step {'hipGraph replays' if args.mode == 'graph' else 'eager blocks'} "
                             f"(whole multiples of --steps) after {args.warmup} warm-up steps and one calibration pass"},
    }
    if not valid:
        out["valid"] = False
        out["invalid_reason"] = f"MGX_LIBMGX={_lib.LIB_PATH}: not the product library (profiling / experiment build)"
    if rank == 0:
        if P == 1:
            out["roofline"] = step_roofline(name, spec, B, m["event_ms"] / S)
        else:                                                   # (--sub-shards: the literal per-launch number stays the `frac`)
            out["roofline"] = step_roofline(name, spec, B // P, sub_shard_launch_ms(env, P, device))
            out["roofline"]["step_frac"] = round(B * A * spec.bytes_step() / (m["event_ms"] / S * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if P == 1 and hint > 1 and args.mode == "graph" and not args.no_pipelined:   # the pipelined variant beside the lock-step headline (all ranks)
        m2, pp = pipelined_point(name, spec, env, args.steps, device, barrier, 99 + rank, hint, min_region_ms=25.0,
                                 agree=lambda n: int(all_max(float(n))))
        pp["value"] = round(G * A * m2["timed_steps"] / all_max(m2["wall_s"]))
        if rank == 0:
            out["pipelined"] = pp
    if world > 1:               # the OTHER scaling point of the same configuration, same run (all ranks): `weak` beside a strong
                                # headline (the default), `strong` beside a weak one
        other = "weak" if args.scaling == "strong" else "strong"
        Go = G0 * world if other == "weak" else G0
        f2, B2 = shard_range(Go, rank, world)
        wl2 = workloads.make(name, batch=B2, first_env=f2, global_batch=Go)
        env2 = wl2.make_env(device, auto_reset=AUTO_RESET)
        m3 = measure_steps(env2, args.steps, args.warmup, args.mode, barrier, seed=4321 + rank,
                           agree=lambda n: int(all_max(float(n))), sub_shards=1)
        env2.check_errors()
        w3max, w3min, S3 = all_max(m3["wall_s"]), -all_max(-m3["wall_s"]), m3["timed_steps"]
        if rank == 0:
            out[other] = {"global_batch": Go, "batch_per_gpu": B2, "value": round(Go * A * S3 / w3max), "unit": "agent-steps/s",
                          "scaling": other, "ms_per_step": round(w3max * 1e3 / S3, 6),
                          "ms_per_step_ranks": {"min": round(w3min * 1e3 / S3, 6), "max": round(w3max * 1e3 / S3, 6)},
                          "timed_steps": S3, "launch": env2.backend.launch_info(B2),
                          "roofline": step_roofline(name, spec, B2, m3["event_ms"] / S3),
                          "note": ("the configuration's batch on EVERY GPU (per-GPU work fixed: flat by construction, no rank waits "
                                   "for another), lock step, measured in this run after the headline" if other == "weak" else
                                   "the configuration's batch split over the GPUs (BASELINE.json configs[3] read literally), "
                                   "lock step, measured in this run after the headline: at an N-th of the batch a launch is a "
                                   "lone wavefront's instruction chain + the launch boundary, not throughput (DESIGN.md §6)")}
        del env2
    if rank == 0:
        out["roofline"]["cache_resident"] = bool(B * A * spec.bytes_step() < 200e6)
        if B * A * spec.bytes_step() < 200e6:
            out["roofline"]["note"] = ("working set fits the 256 MiB Infinity Cache at this batch: see roofline_large for "
                                       "the HBM-resident regime")
        if not args.no_extras and world == 1:                  # everything else: N=1 only
            del env
            torch.cuda.empty_cache()
            out["configs"] = {c: config_point(c, device, 256, 50) for c in ("c2", "c3", "c5") if c != name}
            try:
                out["configs"]["c1"] = c1_point(device)
            except Exception as e:                          # (the dict API must not take the bench line down)
                out["configs"]["c1"] = {"error": repr(e)[:200]}
            if name != "c5":      # C5 is stepped on COMPACT cells (include/mgx.h: MgxCell8); the same workload on the 16-bit cells beside it
                out["configs"]["c5_wide_cells"] = config_point("c5", device, 256, 50, cell_bytes=2)
            # (round 6) C5 with one-hot observations -- the reference's default RL path -- on compact and on 16-bit cells
            for key, cb in (("c5_one_hot", None), ("c5_one_hot_wide_cells", 2)):
                try:
                    out["configs"][key] = one_hot_config_point("c5", device, cell_bytes=cb)
                except Exception as e:
                    out["configs"][key] = {"error": repr(e)[:200]}
            out["configs"]["c3_device_generated"] = config_point("c3", device, 256, 50, device_generated=True)
            out["configs"]["c3_device_generated_steady"] = config_point("c3", device, 256, 50, device_generated=True, steady=True)
            try:
                out["configs"]["c3_short_episodes"] = short_episode_point(device)
            except Exception as e:
                out["configs"]["c3_short_episodes"] = {"error": repr(e)[:300]}
            out["eager"] = {c: eager_point(workloads.make(c), device) for c in ("c4", "c2")}
            # (round 6: sub_shards="auto" answers 1 for per-step calls from Python -- the explicit two chains stay measured here)
            out["eager"]["c4_chains"] = eager_point(workloads.make("c4"), device, sub_shards=2)
            out["eager"]["c4_chains"]["auto_eager"] = 1
            out["fused_rollout"] = rollout_point(workloads.make("c2"), device, 1000)
            # (round 6) the RESIDENT form of C4's shape: 64 view slots per wavefront, the envs' tiles in LDS for the whole launch -- at
            # the largest batch that is resident in one round (49152 envs, 12 wavefronts per CU) and at C4's own 65536 (1.33 rounds)
            out["fused_rollout_c4"] = {str(b): rollout_point(workloads.make("c4", batch=b, global_batch=65536), device, 96, obs_budget=4 << 30)
                                       for b in (49152, 65536)}
            out.update(large_batch_points(device, args.large_batch))
            out["device_generation"] = generation_point(device)
            out["roofline"]["hbm_resident"] = {k: out["roofline_large"][k] for k in ("batch", "frac", "achieved", "ms_per_launch")}
            out["persistent"] = {}
            # (c4: round 6 -- the whole configuration resident, two slices of 16 envs per wavefront; the launch owns every CU's LDS)
            for pname, pbase, pbatch in (("c2", "c2", 4096), ("c4_share8", "c4", 8192), ("c3", "c3", 16384), ("c4", "c4", 65536)):
                try:
                    out["persistent"][pname] = persistent_point(pname, pbase, pbatch, device, T=200 if pbatch > 16384 else 400)
                except Exception as e:                     # (a hand-shake that timed out must not take the bench line down)
                    out["persistent"][pname] = {"error": repr(e)[:200]}
            out["byte_grid_overhead"] = byte_grid_overhead(workloads.make("c4"), device)
            try:                                            # policy in the loop (examples/closed_loop.py), one hipGraph per iteration
                sys.path.insert(0, os.path.join(ROOT, "examples"))
                import closed_loop
                out["closed_loop"] = {c: closed_loop.run(c, steps=100, device=str(device)) for c in ("c4", "c2")}
                # ... and as the double-buffered actor loop: 4 blocks of the batch, each its own [policy -> step] chain
                out["closed_loop"]["c4_4_blocks"] = closed_loop.run("c4", steps=100, device=str(device), sub_shards=4)
            except Exception as e:                          # an example must not take the bench line down
                out["closed_loop"] = {"error": repr(e)[:200]}
            from oracle import binding as ob
            out["cpu_baseline"] = cpu_baseline(wl, ob.max_threads(), 8.0, wl.batch)
            out["cpu_baseline_1core"] = cpu_baseline(wl, 1, 6.0, 2048)
            out["cpu_baseline_python_1core"] = cpu_baseline_python(wl)
        print(json.dumps(out), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
