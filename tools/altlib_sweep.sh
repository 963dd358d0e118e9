#!/bin/bash
# time the fused kernels with each alternative build in altlib/*.so (tuning experiments; GPU box)
cp multigrid_amd/lib/libmgx.so /tmp/libmgx_main.so
echo "main: $(python tools/quick_time.py "$@" 2>&1 | tail -1)"
for L in altlib/*.so; do
  cp $L multigrid_amd/lib/libmgx.so
  echo "$(basename $L): $(python tools/quick_time.py "$@" 2>&1 | tail -1)"
done
cp /tmp/libmgx_main.so multigrid_amd/lib/libmgx.so
