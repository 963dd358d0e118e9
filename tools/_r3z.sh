export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python tools/fuzz_parity.py 240 31 2>&1 | grep -v amdgpu | tail -n 4 > gpurun_out/r3z_fuzz.txt
MGX_LIBMGX=$PWD/multigrid_amd/lib/libmgx_chk.so timeout 400 python tools/fuzz_parity.py 200 77 2>&1 | grep -v amdgpu | tail -n 4 >> gpurun_out/r3z_fuzz.txt
cat gpurun_out/r3z_fuzz.txt | cut -c1-300
