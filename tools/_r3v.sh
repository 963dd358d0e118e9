export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3v.txt
for rep in 1 2; do
for P in "altlib/spans_ref.so c4 65536" "altlib/sp_fix_c4.so c4 65536" "altlib/spans_ref.so c3 16384" "altlib/sp_fix_c3.so c3 16384"; do
    set -- $P
    echo "== $1 $2 $3" >> gpurun_out/r3v.txt
    MGX_LIBMGX=$PWD/$1 MGX_WORKLOAD=$2 MGX_GRAPH=1 timeout 300 python tools/span_probe.py $3 2>&1 | grep -E "wave durations|none of|^B=" >> gpurun_out/r3v.txt
    MGX_LIBMGX=$PWD/$1 MGX_WORKLOAD=$2 MGX_SHOW_STEPS=0 timeout 120 python tools/chain_overlap.py $3 1 2>&1 | grep -E "graph replay|launch duration|gap" >> gpurun_out/r3v.txt
done
done
cat gpurun_out/r3v.txt | cut -c1-200
