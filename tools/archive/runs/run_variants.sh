#!/bin/bash
# A/B runs of the C++ driver (tools/nd_drive.cpp) against variant builds of the library (tools/build/v_*/) and environment knobs.
# usage: bash tools/run_variants.sh [n = 1000] [table = 0/1]
n=${1:-1000}
D=tools/build/nd_drive
[ "${2:-0}" = 1 ] && export ND_DRIVE_TABLE=1
run() { echo "=== $1"; shift; env "$@" timeout 120 $D $n 300 3 -1 0 2>&1 | grep -E "persist 0|levels 5|error|HIP"; }
for rep in 1 2 3; do
run "default (XCD-aware subtree order in the tier)" X=1
run "LS_ND_XCD_TIER=0" LS_ND_XCD_TIER=0
done
