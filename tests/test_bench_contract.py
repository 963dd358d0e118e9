"""bench.py prints ONE JSON line with the fields the driver reads (GPU box only: there is no CPU path to benchmark)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_prints_the_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "40", "--warmup", "5",
                          "--no-extras"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "strong" and d["vs_baseline"] is None    # (N = 1: no other point)
    assert "weak" not in d and "strong" not in d
    assert d["dtype"] == "u8" and d["data"] == "synthetic" and "workload" in d["config"]
    assert "valid" not in d                                     # the product library, no debug knobs
    # the headline is the north-star configuration: Empty-16x16, 4 agents, 65536 envs, all on this GPU
    assert d["config"]["global_batch"] == 65536 and d["config"]["batch_per_gpu"] == 65536 and "16x16" in d["config"]["workload"]
    assert d["timed_steps"] % 40 == 0 and d["timed_region_ms"] >= 45.0
    assert d["value"] > 1e8 and abs(d["value"] - 65536 * 4 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the headline is the lock-step path -- ONE launch of the whole batch per step -- so `roofline.frac` is the literal
    # per-launch number a kernel trace reproduces; the sub-sharded (pipelined) variant is carried beside it with its own
    # per-launch fraction and the aggregate over the step
    assert d["config"]["sub_shards"] == 1 and r["algorithmic_bytes"] == 65536 * 4 * 339
    assert abs(r["ms_per_launch"] - d["ms_per_step"]) / d["ms_per_step"] < 0.1
    p = d["pipelined"]
    assert p["sub_shards"] == 2 and p["launch"]["batch"] * 2 == 65536          # (mgx_sub_shards' answer: two chains, round 5)
    assert p["value"] > d["value"] * 0.9 and p["launch"]["frac"] < p["step_frac"]
    assert abs(p["step_frac"] - p["launch"]["frac"] * p["mean_launches_in_flight"]) < 0.02
