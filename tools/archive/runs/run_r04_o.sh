#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_o; rm -rf $O; mkdir -p $O
timeout 200 tools/build/level_shape > $O/level_shape.txt 2>&1
cat $O/level_shape.txt
