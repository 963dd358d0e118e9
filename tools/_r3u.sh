export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3u.txt
for rep in 1 2; do
for L in altlib/spans_ref.so altlib/sp_fix_c2.so; do
    echo "== $L c2 4096" >> gpurun_out/r3u.txt
    MGX_LIBMGX=$PWD/$L MGX_WORKLOAD=c2 MGX_GRAPH=1 timeout 300 python tools/span_probe.py 4096 2>&1 | grep -E "wave durations|none of|^B=" >> gpurun_out/r3u.txt
    MGX_LIBMGX=$PWD/$L MGX_WORKLOAD=c2 MGX_SHOW_STEPS=0 timeout 120 python tools/chain_overlap.py 4096 1 2>&1 | grep -E "graph replay|launch duration|gap" >> gpurun_out/r3u.txt
done
done
cat gpurun_out/r3u.txt | cut -c1-200
