export TMPDIR=/tmp
mkdir -p gpurun_out
TS=$PWD/multigrid_amd/lib/libmgx_ts.so; SP=$PWD/multigrid_amd/lib/libmgx_spans.so; DBG=$PWD/multigrid_amd/lib/libmgx_dbg.so
MGX_LIBMGX=$SP MGX_WORKLOAD=c4 MGX_GRAPH=1 timeout 300 python tools/span_probe.py 65536 16384 2>&1 | grep -v amdgpu > gpurun_out/r3i_span_c4.txt
MGX_LIBMGX=$SP MGX_WORKLOAD=c2 MGX_GRAPH=1 timeout 300 python tools/span_probe.py 4096 2>&1 | grep -v amdgpu > gpurun_out/r3i_span_c2.txt
MGX_LIBMGX=$TS MGX_WORKLOAD=c4 timeout 300 python tools/stamp_probe.py 65536 1048576 2>&1 | grep -v amdgpu > gpurun_out/r3i_stamps_c4.txt
for S in 0 1024 1; do
  echo "skip=$S c5: $(MGX_LIBMGX=$DBG MGX_WORKLOAD=c5 MGX_SKIP=$S timeout 200 python tools/quick_time.py 32768 2>&1 | grep -v amdgpu)" >> gpurun_out/r3i_halftile.txt
  echo "skip=$S c4: $(MGX_LIBMGX=$DBG MGX_WORKLOAD=c4 MGX_SKIP=$S timeout 200 python tools/quick_time.py 65536 1048576 2>&1 | grep -v amdgpu)" >> gpurun_out/r3i_halftile.txt
done
cat gpurun_out/r3i_span_c4.txt gpurun_out/r3i_halftile.txt; cut -c1-400 gpurun_out/r3i_stamps_c4.txt
