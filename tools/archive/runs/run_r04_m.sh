#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_m; rm -rf $O; mkdir -p $O
LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py cfg4_plane1m 5 2>&1 | grep -E "constructor|nd_plan\]|ls_direct_factor|ls_direct_create" | grep -v "round " > $O/constructor_tail.txt
tail -34 $O/constructor_tail.txt
