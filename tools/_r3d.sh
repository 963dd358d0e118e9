export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3d_gputest.log
SP=$PWD/multigrid_amd/lib/libmgx_spans.so
MGX_LIBMGX=$SP MGX_WORKLOAD=c4 timeout 300 python tools/chain_overlap.py 65536 1 2 4 > gpurun_out/r3_chain_overlap.txt 2>&1
timeout 300 python tools/host_cost.py > gpurun_out/r3d_host_cost.txt 2>&1
for L in multigrid_amd/lib/libmgx.so altlib/obsc17.so altlib/obsc19.so; do
  for W in c2 c3 c4 c5; do
    echo "$(basename $L) $W: $(MGX_LIBMGX=$PWD/$L timeout 200 python bench.py --no-extras --workload $W 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"], d.get("pipelined",{}).get("ms_per_step"))')" >> gpurun_out/r3d_variants.txt
  done
done
cat gpurun_out/r3d_variants.txt; cat gpurun_out/r3d_host_cost.txt; tail -4 gpurun_out/r3d_gputest.log
