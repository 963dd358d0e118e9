#!/bin/bash
# A/B library: lib/libmgx_ab.so = the product library's objects with the 7x7 instantiation unit recompiled with extra defines
#   tools/build_ab.sh -DMGX_ROLL_LAUNDER=0 ...      then   MGX_LIBMGX=multigrid_amd/lib/libmgx_ab.so python tools/...
set -e
cd "$(dirname "$0")/.."
OBJ=build/libmgx.so.obj
mkdir -p build/ab
OUT=${AB_OUT:-multigrid_amd/lib/libmgx_ab.so}          # AB_OUT=<path>: several variants side by side
TAG=$(basename "$OUT" .so)
VS=${AB_V:-7}                                          # AB_V="7 9": the view-size units to recompile (default: 7x7 only)
SKIP=""
NEW=""
for V in $VS; do
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -DMGX_INST_V=$V "$@" -Iinclude -c multigrid_amd/csrc/mgx_fused_inst.hip -o build/ab/${TAG}_v$V.o &
    SKIP="$SKIP -e mgx_fused_v$V.o"
    NEW="$NEW build/ab/${TAG}_v$V.o"
done
wait
OBJS=$(ls $OBJ/*.o | grep -v $SKIP)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $NEW -o "$OUT"
echo built "$OUT" with "$@" "(units: $VS)"
