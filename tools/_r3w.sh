export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3w.txt
for rep in 1 2; do
for L in altlib/ref_v9.so altlib/fix_c5.so; do
  echo "$L c5: $(MGX_LIBMGX=$PWD/$L timeout 200 python bench.py --no-extras --workload c5 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d.get("pipelined",{}).get("ms_per_step"))')" >> gpurun_out/r3w.txt
  echo "$L c5 quick: $(MGX_LIBMGX=$PWD/$L MGX_WORKLOAD=c5 timeout 200 python tools/quick_time.py 32768 2>&1 | grep -v amdgpu)" >> gpurun_out/r3w.txt
done
done
cat gpurun_out/r3w.txt
