cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/genprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/genprof -o g -- python tools/_genprof.py > gpurun_out/genprof.log 2>&1
grep -v "amdgpu.ids\|simple_timer" gpurun_out/genprof.log | tail -8
python - <<EOF
import csv,glob
for f in glob.glob("gpurun_out/genprof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:120], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"])
EOF
