export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3f_ab.txt
for rep in 1 2; do
for L in multigrid_amd/lib/libmgx.so altlib/pre_grp.so; do
  for W in c2 c3; do
    echo "$(basename $L) $W: $(MGX_LIBMGX=$PWD/$L timeout 200 python bench.py --no-extras --workload $W 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d["config"]["launch"])')" >> gpurun_out/r3f_ab.txt
  done
done
done
cat gpurun_out/r3f_ab.txt
