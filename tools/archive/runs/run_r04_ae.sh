#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ae; rm -rf $O; mkdir -p $O
timeout 120 tools/build/spd_inverse 2>&1 | tee $O/spd_inverse.txt
