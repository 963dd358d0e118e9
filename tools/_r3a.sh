export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r3a_gputest.log
TS=$PWD/multigrid_amd/lib/libmgx_ts.so; DBG=$PWD/multigrid_amd/lib/libmgx_dbg.so
MGX_LIBMGX=$TS MGX_WORKLOAD=c4 timeout 300 python tools/chain_overlap.py 65536 1 2 4 > gpurun_out/r3_chain_overlap.txt 2>&1
MGX_LIBMGX=$TS MGX_WORKLOAD=c4 MGX_GRAPH=1 timeout 300 python tools/span_probe.py 65536 8192 > gpurun_out/r3a_span_c4.txt 2>&1
MGX_LIBMGX=$TS MGX_WORKLOAD=c2 MGX_GRAPH=1 timeout 300 python tools/span_probe.py 4096 > gpurun_out/r3a_span_c2.txt 2>&1
MGX_LIBMGX=$DBG MGX_WORKLOAD=c2 MGX_GS=0,1,2,4 MGX_WPBS=1,2,4 timeout 400 python tools/graph_g.py 4096 8192 > gpurun_out/r3a_g_c2.txt 2>&1
MGX_LIBMGX=$DBG MGX_WORKLOAD=c3 MGX_GS=0,2,4,8 MGX_WPBS=2,4 timeout 300 python tools/graph_g.py 16384 > gpurun_out/r3a_g_c3.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
tail -5 gpurun_out/r3a_gputest.log
