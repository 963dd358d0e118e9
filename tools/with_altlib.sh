#!/bin/bash
# run a command with an alternative build of the library in place: tools/with_altlib.sh <alt.so> <command...>
ALT=$1; shift
cp multigrid_amd/lib/libmgx.so /tmp/libmgx_main.so
cp $ALT multigrid_amd/lib/libmgx.so
"$@"
cp /tmp/libmgx_main.so multigrid_amd/lib/libmgx.so
