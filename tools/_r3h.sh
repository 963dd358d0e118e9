export TMPDIR=/tmp
mkdir -p gpurun_out
TS=$PWD/multigrid_amd/lib/libmgx_ts.so; SP=$PWD/multigrid_amd/lib/libmgx_spans.so
MGX_LIBMGX=$TS MGX_WORKLOAD=c4 timeout 300 python tools/stamp_probe.py 65536 1048576 2>&1 | grep -v amdgpu > gpurun_out/r3h_stamps_c4.txt
MGX_LIBMGX=$TS MGX_WORKLOAD=c4 MGX_GRAPH=1 timeout 300 python tools/span_probe.py 65536 2>&1 | grep -v amdgpu > gpurun_out/r3h_span_c4.txt
MGX_LIBMGX=$TS MGX_WORKLOAD=c5 timeout 300 python tools/stamp_probe.py 32768 2>&1 | grep -v amdgpu > gpurun_out/r3h_stamps_c5.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3h_bench.json 2> gpurun_out/r3h_bench.err
timeout 1500 python tools/profile_round.py r3 > gpurun_out/r3h_profile.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3h_gputest.log
tail -3 gpurun_out/r3h_gputest.log; head -c 1500 gpurun_out/r3h_bench.json; cat gpurun_out/r3h_stamps_c4.txt
