#!/bin/bash
# Round profile (GPU box): rocprofv3 kernel-trace stats of the default bench command, plus PMC passes of the fused
# kernel at the bench batch and at the large (HBM-resident) batch.  Writes gpurun_out/<tag>/..., summaries are then
# copied to profiles/ by hand.      usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r1}; OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -o bench -- python bench.py --no-extras --settle-steps 0 > $OUT/bench_trace.log 2>&1
grep '^{"metric"' $OUT/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
tools/pmc_probe.sh $OUT/pmc_b4096 4096 > $OUT/pmc_b4096.log 2>&1
tools/pmc_probe.sh $OUT/pmc_b1m 1048576 > $OUT/pmc_b1m.log 2>&1
python - <<PY
import csv, glob, json, os
out = "$OUT"
rows = []
for f in glob.glob(os.path.join(out, "bench_trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(r)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(os.path.join(out, "bench_kernel_stats.txt"), "w") as fh:
    fh.write("# rocprofv3 --kernel-trace --stats -- python bench.py --no-extras --settle-steps 0\n")
    for r in rows[:8]:
        fh.write(f"{r['Name'][:90]:90s} calls={r['Calls']:>6s} avg_ns={float(r['AverageNs']):10.1f} total_ns={r['TotalDurationNs']} pct={r['Percentage']}\n")
print(open(os.path.join(out, "bench_kernel_stats.txt")).read())
PY
python tools/traffic_from_pmc.py $OUT $TAG
cat $OUT/pmc_b4096/summary.txt | grep -i "kernel_stats\|FETCH\|WRITE\|GRBM\|VALU\|SALU\|WAVE_CYCLES\|WAIT_ANY\|TCC"
echo ----
cat $OUT/pmc_b1m/summary.txt | grep -i "kernel_stats\|FETCH\|WRITE\|GRBM\|VALU\|SALU\|WAVE_CYCLES\|WAIT_ANY\|TCC"
