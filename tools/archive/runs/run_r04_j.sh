#!/bin/bash
# pass J: fp64 instruction rates (tools/ubench/fp64_rate.hip), page faults per stage of the constructor
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_j; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/build/fp64_rate > $O/fp64_rate.txt 2>&1
cat $O/fp64_rate.txt
LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py cfg4_plane1m 8 2>&1 | grep -E "constructor|nd_plan\]|ls_direct_factor" | grep -v "round " > $O/constructor_faults.txt
tail -60 $O/constructor_faults.txt
