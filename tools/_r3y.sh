export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3y.txt
MGX_LIBMGX=$PWD/altlib/direct1.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "golden or random_states" 2>&1 | tail -4 > gpurun_out/r3y_parity.log
for rep in 1 2; do
for L in multigrid_amd/lib/libmgx.so altlib/direct1.so; do
for W in c2 c3 c4; do
  echo "$L $W: $(MGX_LIBMGX=$PWD/$L timeout 200 python bench.py --no-extras --workload $W 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d.get("pipelined",{}).get("ms_per_step"))')" >> gpurun_out/r3y.txt
done
echo "$L: $(MGX_LIBMGX=$PWD/$L MGX_WORKLOAD=c4 timeout 200 python tools/quick_time.py 65536 1048576 2>&1 | grep -v amdgpu)" >> gpurun_out/r3y.txt
done
done
cat gpurun_out/r3y_parity.log gpurun_out/r3y.txt
