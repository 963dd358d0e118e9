export TMPDIR=/tmp
mkdir -p gpurun_out
DBG=$PWD/multigrid_amd/lib/libmgx_dbg.so
MGX_LIBMGX=$DBG MGX_WORKLOAD=c4 timeout 600 python tools/stagger_probe.py 65536 2>&1 | grep -v amdgpu > gpurun_out/r3j_stagger_c4.txt
MGX_LIBMGX=$DBG MGX_WORKLOAD=c5 MGX_STAGGER=0,8,16,32 timeout 600 python tools/stagger_probe.py 32768 2>&1 | grep -v amdgpu > gpurun_out/r3j_stagger_c5.txt
cat gpurun_out/r3j_stagger_c4.txt gpurun_out/r3j_stagger_c5.txt
