export TMPDIR=/tmp
mkdir -p gpurun_out; rm -f gpurun_out/r3m_ab.txt
for rep in 1 2; do
for L in altlib/spans_old.so multigrid_amd/lib/libmgx_spans.so; do
  for W in "c2 4096" "c3 16384" "c4 16384"; do
    set -- $W
    echo "== $L $1 $2" >> gpurun_out/r3m_ab.txt
    MGX_LIBMGX=$PWD/$L MGX_WORKLOAD=$1 MGX_GRAPH=1 timeout 300 python tools/span_probe.py $2 2>&1 | grep -E "wave durations|took|none of|^B=" >> gpurun_out/r3m_ab.txt
  done
done
done
cat gpurun_out/r3m_ab.txt
