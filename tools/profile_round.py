#!/usr/bin/env python3
"""Round profile (GPU box): for every BASELINE configuration and for the HBM-resident (1M-env) points, WARM rocprofv3
kernel-trace durations (the first `warm` launches of each kernel are discarded) and the HBM traffic from separate --pmc
passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains).  Also a kernel-trace --stats of the default bench
command.  Writes gpurun_out/prof_<tag>/{summary.txt, traffic.json, bench_kernel_stats.txt}; copy them to profiles/.

    python tools/profile_round.py <tag> [--quick]
"""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
OUT = os.path.join(ROOT, "gpurun_out", f"prof_{TAG}")
os.makedirs(OUT, exist_ok=True)
ENV = dict(os.environ, TMPDIR="/tmp")
WARM, N = 40, 60
WARM_LARGE = 250      # the 1M-env point: steady state of the rollout (tools/time_curve.py)

#: (key, MGX_WORKLOAD, batch, extra env)
POINTS = [("c2", "c2", 4096, {}), ("c3", "c3", 16384, {}), ("c4", "c4", 65536, {}), ("c5", "c5", 32768, {}),
          ("c5_wide", "c5", 32768, {"MGX_CELL_BYTES": "2"}),      # C5 on the 16-bit cells (its own format is the compact one)
          ("c4_share8", "c4", 8192, {}), ("c4_part", "c4", 32768, {}), ("c5_part", "c5", 16384, {}),      # the sub-shard launches of bench.py (two chains)
          ("large", "c4", 1 << 20, {"MGX_ONE_HOT_STEP": "1"})]


def kind_of(name: str):
    m = re.search(r"mgx_obs_kernel<([^>]*)>", name)
    if m:                                   # <V, OH, STREAM, DMA, C8>: gen_obs for views up to 7x7 (its own entry point)
        return "gen_obs" + ("_one_hot" if m.group(1).split(",")[1].strip() == "true" else "")
    m = re.search(r"mgx_fused_kernel<([^>]*)>", name)
    if m:                                   # <V, MODE, HOOKS, AR, OH, GEN, STREAM, DMA, GRP, SHAPE, C8>
        t = [x.strip() for x in m.group(1).split(",")] + ["false"] * 8
        return ({"0": "gen_obs", "1": "step", "2": "rollout"}[t[1]] + ("_one_hot" if t[4] == "true" else "")
                + ("_generate" if t[5] == "true" else ""))
    for k in ("one_hot_kernel", "full_obs_kernel", "reset_done_kernel"):
        if k in name:
            return k
    return None


PARSE_ONLY = "--parse-only" in sys.argv      # re-summarise CSVs that are already there


def rocprof(args, outdir, cmd, extra_env):
    if PARSE_ONLY:
        return
    subprocess.run(["rocprofv3", *args, "--output-format", "csv", "-d", outdir, "-o", "p", "--", *cmd],
                   env=dict(ENV, **extra_env), cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)


def durations(outdir, warm):
    per = collections.defaultdict(list)
    for f in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        for r in rows:
            k = kind_of(r["Kernel_Name"])
            if k:
                per[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: v[warm:] for k, v in per.items() if len(v) > warm}


def counters(outdir):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kind_of(r["Kernel_Name"])
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


lines, traffic = [], {"_comment": "HBM-side traffic per launch from rocprofv3 --pmc passes (tools/profile_round.py; FETCH_SIZE and "
                                  "WRITE_SIZE in separate passes, kernel-trace only).  bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: "
                                  "both counters are in KB and FETCH_SIZE reports half of a wide coalesced read stream on gfx950 "
                                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated per the same guide.",
                      "round": TAG}
quick = "--quick" in sys.argv
ONLY = next((a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")), None)   # --only=c4_part,bench: just these
for key, wl, B, xenv in POINTS:
    if ONLY is not None and key not in ONLY:
        continue
    e = dict(MGX_WORKLOAD=wl, **xenv)
    if B < (1 << 20):
        e["MGX_GRAPH"] = "1"                 # the configurations' steps run as hipGraph replays, as bench.py times them
    warm = WARM_LARGE if B >= (1 << 20) else WARM
    cmd = [sys.executable, "tools/large_step.py", str(B), str(N), str(warm)]
    d = os.path.join(OUT, f"{key}_trace")
    rocprof(["--kernel-trace"], d, cmd, e)
    for k, v in sorted(durations(d, warm).items()):
        lines.append(f"trace {key} B={B} {k}: warm_calls={len(v)} avg_ns={sum(v) / len(v):.1f} min_ns={min(v)} max_ns={max(v)}")
    if quick:
        continue
    cmd = [sys.executable, "tools/large_step.py", str(B), "8", "4"]
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(OUT, f"{key}_{ctr}")
        rocprof(["--kernel-trace", "--pmc", ctr], d, cmd, e)
        for k, cs in counters(d).items():
            v = cs[ctr][2:]                     # (first launches: cold caches)
            if v:
                got.setdefault(k, {})[ctr] = sum(v) / len(v)
    for k, c in sorted(got.items()):
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            by = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
            traffic.setdefault(f"{key}_{k}", {})[str(B)] = {"fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
                                                            "bytes": by}
            lines.append(f"pmc   {key} B={B} {k}: FETCH_SIZE={c['FETCH_SIZE']:.1f} KB WRITE_SIZE={c['WRITE_SIZE']:.1f} KB -> {by} B per launch")

# the default bench command under kernel-trace --stats
d = os.path.join(OUT, "bench_trace")
log = os.path.join(OUT, "bench_under_rocprof.json")
with open(log, "a" if PARSE_ONLY else "w") as fh:
    subprocess.run(["true"] if PARSE_ONLY else ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "-o", "bench", "--",
                    sys.executable, "bench.py", "--no-extras", "--no-pipelined"], env=ENV, cwd=ROOT, stdout=fh, stderr=subprocess.DEVNULL)
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(os.path.join(OUT, "bench_kernel_stats.txt"), "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --no-pipelined   (C4: 65536 envs on one GPU, lock step only:\n"
             "# the sub-sharded variant launches the same kernel at half the batch and would mix into its average)\n")
    for r in rows[:6]:
        fh.write(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):10.1f} min_ns={r['MinNs']} "
                 f"total_ns={r['TotalDurationNs']} pct={r['Percentage']}\n")
txt = "\n".join(lines)
open(os.path.join(OUT, "summary.txt"), "w").write(
    f"# tools/profile_round.py {TAG}: rocprofv3 kernel-trace, first {WARM} launches of each kernel discarded; the steps of the\n"
    f"# configurations (B < 1M) are hipGraph replays, as bench.py times them; separate --pmc passes (FETCH_SIZE, WRITE_SIZE)\n" + txt + "\n")
json.dump(traffic, open(os.path.join(OUT, "traffic_only.json" if ONLY is not None else "traffic.json"), "w"), indent=1)
print(txt)
print(open(os.path.join(OUT, "bench_kernel_stats.txt")).read())
