"""The multi-rank path on the REAL kernels: two `torch.distributed.run` ranks each step their shard of C4 (65536 envs,
strong-scaled: 32768 per rank) on the HIP backend; concat(shards) must equal one process stepping the whole batch.  The
single-GPU test box has one device, so both ranks use device 0 (MGX_BENCH_ONE_GPU=1, gloo for the rendezvous) -- the code
path is the one bench.py --gpus N runs, minus RCCL's barrier.  Also: bench.py itself under 2 ranks prints its line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(args, extra_env=None, timeout=900, nproc=2):
    env = dict(os.environ, MGX_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


@pytest.mark.parametrize("name,G,T,N", [("c4", 65536, 12, 2), ("c3", 16384, 12, 2), ("c5", 2048, 6, 2), ("c4", 65536, 8, 8)])
def test_hip_ranks_equal_single_process(tmp_path, name, G, T, N):
    """N ranks (2, and the 8 of the named 8-GPU target: 8192 envs of C4 each) == one process stepping the whole batch."""
    out = _torchrun([os.path.join(ROOT, "tests", "shard_worker.py"), name, str(G), str(T), str(tmp_path)], nproc=N)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tests.shard_worker import run_shard
    whole = run_shard(name, G, T, 0, G, torch.device("cuda", 0))
    shards = [np.load(os.path.join(tmp_path, f"shard{r}.npz")) for r in range(N)]
    assert int(shards[0]["first"]) == 0 and int(shards[1]["first"]) == int(shards[0]["count"])
    assert sum(int(s["count"]) for s in shards) == G and all(int(s["count"]) == G // N for s in shards)
    for k, v in whole.items():
        cat = np.concatenate([s[k] for s in shards])
        assert cat.tobytes() == v.tobytes(), k


def test_bench_under_two_ranks_prints_the_strong_line_by_default():
    """The default for N > 1 (round 5): BASELINE.json configs[3] read literally -- the configuration's 65 536 envs SPLIT over the
    ranks -- is `value`; the weak point (the configuration's batch on every GPU) rides along as `weak`."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "strong" not in d
    assert d["config"]["global_batch"] == 65536 and d["config"]["batch_per_gpu"] == 32768 and d["config"]["configuration_batch"] == 65536
    assert abs(d["value"] - 65536 * 4 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    assert d["roofline"]["algorithmic_bytes"] == 32768 * 4 * 339
    wk = d["weak"]
    assert wk["scaling"] == "weak" and wk["global_batch"] == 2 * 65536 and wk["batch_per_gpu"] == 65536
    assert wk["roofline"]["algorithmic_bytes"] == 65536 * 4 * 339
    assert abs(wk["value"] - 2 * 65536 * 4 / (wk["ms_per_step"] / 1e3)) / wk["value"] < 0.01
    assert "roofline" in d and "configs" not in d and "cpu_baseline" not in d      # extras are N=1 only


def test_bench_under_two_ranks_weak_line_carries_the_strong_point():
    """`--scaling weak`: every rank steps the configuration's 65 536 envs (`value` = all ranks' agent-steps over the slowest
    rank's time), and the same run measures the configuration's batch split over the ranks as `strong`."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--scaling", "weak"])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "weak" not in d
    assert d["config"]["global_batch"] == 2 * 65536 and d["config"]["batch_per_gpu"] == 65536 and d["config"]["configuration_batch"] == 65536
    assert abs(d["value"] - 2 * 65536 * 4 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    assert d["roofline"]["algorithmic_bytes"] == 65536 * 4 * 339
    st = d["strong"]
    assert st["global_batch"] == 65536 and st["batch_per_gpu"] == 32768 and st["roofline"]["algorithmic_bytes"] == 32768 * 4 * 339
    assert abs(st["value"] - 65536 * 4 / (st["ms_per_step"] / 1e3)) / st["value"] < 0.01
    assert "configs" not in d and "cpu_baseline" not in d


def test_bench_under_eight_ranks_prints_the_8_gpu_line():
    """`bench.py --gpus 8` as the driver launches it (one rank per GPU; here all eight on device 0): the first real SCALE run
    must not fail on plumbing.  The headline is the shape the north star's 1e8 agent-steps/s target is stated on: the 65 536 envs
    of C4 split over the eight ranks (8192 each, the shape-specialised latency kernel); 65 536 envs on every rank ride along
    as `weak`."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], nproc=8, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong"
    assert d["config"]["global_batch"] == 65536 and d["config"]["batch_per_gpu"] == 8192 and d["config"]["sub_shards"] == 1
    assert d["config"]["launch"]["fixed_shape"] == 1
    assert "roofline" in d and d["roofline"]["algorithmic_bytes"] == 8192 * 4 * 339
    assert abs(d["value"] - 65536 * 4 / (d["ms_per_step"] / 1e3)) / d["value"] < 0.01
    wk = d["weak"]                                                          # the configuration's batch on every GPU, same run
    assert wk["global_batch"] == 8 * 65536 and wk["batch_per_gpu"] == 65536 and wk["roofline"]["algorithmic_bytes"] == 65536 * 4 * 339
    assert abs(wk["value"] - 8 * 65536 * 4 / (wk["ms_per_step"] / 1e3)) / wk["value"] < 0.01


def test_plain_bench_gpus_8_launches_itself():
    """`python bench.py --gpus 8` WITHOUT a launcher (how the driver's 1-GPU BENCH command would read with N swapped in): bench.py
    becomes the launcher (torch.distributed.run, 127.0.0.1) and rank 0 prints the same one line."""
    env = dict(os.environ, MGX_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["batch_per_gpu"] == 8192 and d["scaling"] == "strong" and d["weak"]["batch_per_gpu"] == 65536
    assert d["ms_per_step_ranks"]["min"] <= d["ms_per_step_ranks"]["max"] == d["ms_per_step"]
