#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_m; rm -rf $O; mkdir -p $O
LS_PLAN_TIMING=1 timeout 600 python tools/bench_remesh.py cfg3_dragon250k 100 4 2>&1 | grep -E "nd_plan\]|ls_direct|rebuild|constructor" | grep -v "round " | tail -60 | tee $O/remesh_timing.txt
