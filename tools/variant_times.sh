#!/bin/bash
# step / gen_obs kernel time of the product library and of every altlib/*.so variant (GPU box)
#   usage: MGX_WORKLOAD=c4 tools/variant_times.sh 65536 1048576
for L in multigrid_amd/lib/libmgx.so $(ls altlib/*.so 2>/dev/null); do
  echo "$(basename $L): $(MGX_LIBMGX=$PWD/$L python tools/quick_time.py "$@" 2>&1 | tail -1)"
done
