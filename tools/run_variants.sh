#!/bin/bash
# A/B runs of the C++ driver (tools/nd_drive.cpp) against variant builds of the library (tools/build/v_*/) and environment knobs.
# usage: bash tools/run_variants.sh [n = 1000] [table = 0/1]
n=${1:-1000}
D=tools/build/nd_drive
[ "${2:-0}" = 1 ] && export ND_DRIVE_TABLE=1
run() { echo "=== $1"; shift; env "$@" timeout 120 $D $n 200 3 2>&1 | awk '/persist 0:/{print; exit} {print}' | grep -E "persist 0|levels|error|HIP"; }
run "default" X=1
run "round-2 rule (LS_ND_INFLIGHT=6, 4 waves)" LS_ND_INFLIGHT=6 LS_ND_BW_LONG=100000
run "LS_ND_TILES=500" LS_ND_TILES=500
run "LS_ND_TILES=2000" LS_ND_TILES=2000
run "LS_ND_BW_LONG=900" LS_ND_BW_LONG=900
run "LS_ND_BW_LONG=400" LS_ND_BW_LONG=400
for v in tools/build/v_*; do run "$(basename $v)" LD_LIBRARY_PATH=$v; done
