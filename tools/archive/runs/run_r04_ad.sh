#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ad; rm -rf $O; mkdir -p $O
( timeout 500 python tools/soak_threads.py 150; timeout 500 python tools/soak_threads.py 150 ) 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/soak_threads.txt
