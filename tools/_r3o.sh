export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3o_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3o_parity.log
for rep in 1 2; do
for L in altlib/spans_old.so multigrid_amd/lib/libmgx_spans.so; do
  for W in "c2 4096" "c3 16384" "c4 16384"; do
    set -- $W
    echo "== $L $1 $2" >> gpurun_out/r3o_ab.txt
    MGX_LIBMGX=$PWD/$L MGX_WORKLOAD=$1 MGX_GRAPH=1 timeout 300 python tools/span_probe.py $2 2>&1 | grep -E "wave durations|took|none of|^B=" >> gpurun_out/r3o_ab.txt
  done
done
done
for rep in 1 2; do
for W in c2 c3 c4 c5; do
  echo "$W: $(timeout 200 python bench.py --no-extras --workload $W 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])')" >> gpurun_out/r3o_ab.txt
done
done
cat gpurun_out/r3o_parity.log; grep -E "==|all |fallback|none|^c[0-9]" gpurun_out/r3o_ab.txt
