export TMPDIR=/tmp
mkdir -p gpurun_out
SP=$PWD/multigrid_amd/lib/libmgx_spans.so; DBG=$PWD/multigrid_amd/lib/libmgx_dbg.so
MGX_LIBMGX=$SP MGX_WORKLOAD=c4 timeout 300 python tools/chain_overlap.py 65536 1 2 4 > gpurun_out/r3_chain_overlap.txt 2>&1
MGX_LIBMGX=$DBG MGX_WORKLOAD=c2 timeout 600 python tools/group_sweep.py 1024 2048 4096 8192 16384 > gpurun_out/r3b_groups_c2.txt 2>&1
MGX_LIBMGX=$DBG MGX_WORKLOAD=c3 timeout 600 python tools/group_sweep.py 4096 8192 16384 > gpurun_out/r3b_groups_c3.txt 2>&1
tail -3 gpurun_out/r3b_groups_c2.txt gpurun_out/r3b_groups_c3.txt
