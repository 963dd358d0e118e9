#!/bin/bash
# Dynamic per-wavefront counters of the fused kernels with one phase skipped at a time (GPU box; needs the tools' build:
# python -m multigrid_amd.build --debug-knobs).
# usage: [COUNTERS="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"] [MASKS="0 1 2 4 8 16 32 63"] tools/valu_by_phase.sh <outdir> [batch]
set -u
OUT=${1:-gpurun_out/valu}; B=${2:-262144}
COUNTERS=${COUNTERS:-"SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES"}
MASKS=${MASKS:-"0 1 2 4 8 16 32 64 128 63"}
export TMPDIR=/tmp
export MGX_LIBMGX=$PWD/multigrid_amd/lib/libmgx_dbg.so
mkdir -p $OUT
for M in $MASKS; do
  MGX_SKIP=$M rocprofv3 --kernel-trace --pmc $COUNTERS --output-format csv \
      -d $OUT/m$M -o p -- python tools/large_step.py $B 4 > $OUT/m$M.log 2>&1 || echo "mask $M failed"
  python - "$OUT/m$M" "$M" <<'PY'
import csv, glob, re, sys, collections
d, m = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mgx_fused" not in k and "mgx_obs_kernel" not in k: continue
        mode = "0" if "mgx_obs_kernel" in k else re.search(r"mgx_fused_kernel<\d+, (\d)", k).group(1)
        name = {"0": "gen_obs", "1": "step", "2": "roll"}[mode]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, c in acc.items():
    w = sum(c["SQ_WAVES"]) / len(c["SQ_WAVES"])
    print(f"skip={m:>3s} {name:8s} per-wave " + " ".join(f"{k.replace('SQ_', '')} {sum(v)/len(v)/w:9.1f}" for k, v in sorted(c.items()) if k != "SQ_WAVES") + f" waves {w:.0f}")
PY
done
