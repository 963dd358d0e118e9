#!/bin/bash
# A/B library: lib/libmgx_ab.so = the product library's objects with the 7x7 instantiation unit recompiled with extra defines
#   tools/build_ab.sh -DMGX_ROLL_LAUNDER=0 ...      then   MGX_LIBMGX=multigrid_amd/lib/libmgx_ab.so python tools/...
set -e
cd "$(dirname "$0")/.."
OBJ=build/libmgx.so.obj
mkdir -p build/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -DMGX_INST_V=7 "$@" -Iinclude -c multigrid_amd/csrc/mgx_fused_inst.hip -o build/ab/mgx_fused_v7.o
OBJS=$(ls $OBJ/*.o | grep -v mgx_fused_v7.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/ab/mgx_fused_v7.o -o multigrid_amd/lib/libmgx_ab.so
echo built multigrid_amd/lib/libmgx_ab.so with "$@"
