export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3g_gputest.log
tail -3 gpurun_out/r3g_gputest.log
timeout 200 python tools/bw_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r3g_bw.txt
timeout 300 python tools/host_cost.py 2>&1 | grep -v amdgpu > gpurun_out/r3g_host_cost.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3g_bench.json 2> gpurun_out/r3g_bench.err
cat gpurun_out/r3g_bw.txt gpurun_out/r3g_host_cost.txt; head -c 600 gpurun_out/r3g_bench.json
