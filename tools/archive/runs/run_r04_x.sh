#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_x; rm -rf $O; mkdir -p $O
for S in 128 0 128 0; do echo "== LS_GEMM_SMALL_TILES=$S"; LS_PLAN_TIMING=1 LS_GEMM_SMALL_TILES=$S timeout 300 python tools/profile_constructor.py cfg4_plane1m 6 2>&1 | grep -E "constructor|nd_plan\]|ls_direct_factor\]|ls_direct_create" | grep -v "round " | tail -22; done > $O/stages.txt
cat $O/stages.txt
