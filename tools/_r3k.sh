export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O2 -Wno-unused-result tools/hwprobe/valu_rate.hip -o tools/hwprobe/valu_rate 2>/dev/null && timeout 300 tools/hwprobe/valu_rate > gpurun_out/r3k_valu_rate.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3k_gputest.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r3k_bench.json 2> gpurun_out/r3k_bench.err
tail -3 gpurun_out/r3k_gputest.log; grep -E "pk_|dot2|mad_i32_i16|med3|mad_i32_i24|v_add_u32 " gpurun_out/r3k_valu_rate.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3k_bench.json').read())
print("c4", d["ms_per_step"], d["roofline"]["frac"], "pipelined", d["pipelined"]["ms_per_step"])
for k,v in d["configs"].items(): print(k, v["ms_per_step"], v["roofline"]["frac"])
print("large", d["roofline_large"]["ms_per_launch"], d["gen_obs_large"]["ms_per_launch"], d["one_hot_large"]["ms_per_launch"])
print("eager", d["eager"]["c2"]["ms_per_step"], d["eager"]["c4"]["ms_per_step"], "rollout", d["fused_rollout"]["ms_per_step"])
PY
