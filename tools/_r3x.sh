export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3x.txt
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size.py tests/test_one_hot_fused.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3x_parity.log
for rep in 1 2; do
for W in c2 c3 c4 c5; do
  echo "$W: $(timeout 200 python bench.py --no-extras --workload $W 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d.get("pipelined",{}).get("ms_per_step"))')" >> gpurun_out/r3x.txt
done
done
MGX_WORKLOAD=c4 timeout 200 python tools/quick_time.py 65536 1048576 2>&1 | grep -v amdgpu >> gpurun_out/r3x.txt
cat gpurun_out/r3x_parity.log gpurun_out/r3x.txt
