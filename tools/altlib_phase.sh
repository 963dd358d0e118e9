#!/bin/bash
# phase_probe with each alternative build in altlib/*.so (tuning experiments; GPU box)
cp multigrid_amd/lib/libmgx.so /tmp/libmgx_main.so
echo "== main"; python tools/${PROBE:-phase_probe.py} "$@" 2>&1 | grep -v amdgpu
for L in altlib/*.so; do
  cp $L multigrid_amd/lib/libmgx.so
  echo "== $(basename $L)"; python tools/${PROBE:-phase_probe.py} "$@" 2>&1 | grep -v amdgpu
done
cp /tmp/libmgx_main.so multigrid_amd/lib/libmgx.so
