#!/bin/bash
# A/B runs of the C++ driver (tools/nd_drive.cpp) against variant builds of the library (tools/build/v_*/) and environment knobs.
# usage: bash tools/run_variants.sh [n = 1000]
n=${1:-1000}
D=tools/build/nd_drive
run() { echo "=== $1"; shift; env "$@" timeout 120 $D $n 200 3 2>&1 | grep -E "persist 0|error|HIP"; }
run "baseline" X=1
run "LS_ND_NO_FUSE_ROOT" LS_ND_NO_FUSE_ROOT=1
run "LS_ND_INFLIGHT=4" LS_ND_INFLIGHT=4
run "LS_ND_INFLIGHT=8" LS_ND_INFLIGHT=8
run "LS_ND_INFLIGHT=12" LS_ND_INFLIGHT=12
run "LS_ND_LONG_UP=64" LS_ND_LONG_UP=64
for v in tools/build/v_*; do run "$(basename $v)" LD_LIBRARY_PATH=$v; done
