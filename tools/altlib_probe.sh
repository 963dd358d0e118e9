#!/bin/bash
# run g_probe with an alternative build of the library (tuning experiments): tools/altlib_probe.sh <alt.so> [batches...]
ALT=$1; shift
cp multigrid_amd/lib/libmgx.so /tmp/libmgx_main.so
cp $ALT multigrid_amd/lib/libmgx.so
python tools/g_probe.py "$@" 2>&1 | grep -v amdgpu.ids
cp /tmp/libmgx_main.so multigrid_amd/lib/libmgx.so
