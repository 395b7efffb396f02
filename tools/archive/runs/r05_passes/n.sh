#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_n; rm -rf $O; mkdir -p $O
for n in 50 100 150 265 500 1000; do ND_DRIVE_GRAPH=1 timeout 200 tools/build/nd_drive $n 500 3 -1 2>&1 | grep -E "replayed" | sed "s/^/[n=$n] /"; done | tee $O/graph_replay.txt
