export TMPDIR=/tmp
mkdir -p gpurun_out
SP=$PWD/multigrid_amd/lib/libmgx_spans.so
MGX_LIBMGX=$SP MGX_WORKLOAD=c4 timeout 300 python tools/chain_overlap.py 65536 1 2 4 > gpurun_out/r3_chain_overlap.txt 2>&1
for L in $SP altlib/obs0.so altlib/obs1.so altlib/obs3.so altlib/obs16.so altlib/obs17.so altlib/obs18.so; do
  echo "=== $L" >> gpurun_out/r3c_obs_policy.txt
  MGX_LIBMGX=$PWD/$( [ "${L:0:1}" = "/" ] && realpath --relative-to=$PWD $L || echo $L ) MGX_WORKLOAD=c4 MGX_SHOW_STEPS=0 timeout 120 python tools/chain_overlap.py 65536 1 2>&1 | grep -E "graph replay|in-kernel|launch duration|gap" >> gpurun_out/r3c_obs_policy.txt
done
cat gpurun_out/r3c_obs_policy.txt | cut -c1-220
