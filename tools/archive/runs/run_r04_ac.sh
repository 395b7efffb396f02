#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ac; rm -rf $O; mkdir -p $O
( timeout 500 python tools/soak_constructor.py 400 1; timeout 500 python tools/soak_constructor.py 400 2 ) 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
