export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3t_policy.txt
for rep in 1 2; do
for L in multigrid_amd/lib/libmgx_spans.so altlib/sp_out2.so altlib/sp_out17.so altlib/sp_out19.so altlib/sp_obs19.so; do
  for W in "c4 65536" "c2 4096"; do
    set -- $W
    echo "=== $L $1 $2" >> gpurun_out/r3t_policy.txt
    MGX_LIBMGX=$PWD/$L MGX_WORKLOAD=$1 MGX_SHOW_STEPS=0 timeout 120 python tools/chain_overlap.py $2 1 2>&1 | grep -E "graph replay|launch duration|gap" >> gpurun_out/r3t_policy.txt
  done
done
done
cat gpurun_out/r3t_policy.txt | cut -c1-210
