export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3s_gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3s_bench.json 2> gpurun_out/r3s_bench.err
timeout 1500 python tools/profile_round.py r3 > gpurun_out/r3s_profile.log 2>&1
SP=$PWD/multigrid_amd/lib/libmgx_spans.so
MGX_LIBMGX=$SP MGX_WORKLOAD=c4 timeout 300 python tools/chain_overlap.py 65536 1 2 4 > gpurun_out/r3_chain_overlap.txt 2>&1
timeout 300 python tools/host_cost.py 2>&1 | grep -v amdgpu > gpurun_out/r3s_host_cost.txt
tail -3 gpurun_out/r3s_gputest.log; tail -12 gpurun_out/r3s_profile.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3s_bench.json').read())
print("c4", d["ms_per_step"], d["roofline"]["frac"], "pipelined", d["pipelined"]["ms_per_step"])
for k,v in d["configs"].items(): print(k, v["ms_per_step"], v["roofline"]["frac"], v.get("pipelined",{}).get("ms_per_step"))
print("large", d["roofline_large"]["ms_per_launch"], d["gen_obs_large"]["ms_per_launch"], d["one_hot_large"]["ms_per_launch"])
print("eager", d["eager"]["c2"]["ms_per_step"], d["eager"]["c4"]["ms_per_step"], "rollout", d["fused_rollout"]["ms_per_step"])
PY
