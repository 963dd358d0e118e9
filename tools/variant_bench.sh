#!/bin/bash
# graph-replay ms_per_step of bench.py --no-extras for every workload, product library vs altlib/*.so (GPU box)
for L in multigrid_amd/lib/libmgx.so $(ls altlib/*.so 2>/dev/null); do
  for W in ${WORKLOADS:-c2 c3 c4 c5}; do
    echo "$(basename $L) $W: $(MGX_LIBMGX=$PWD/$L python bench.py --no-extras --workload $W 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
