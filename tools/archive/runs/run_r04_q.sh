#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_q; rm -rf $O; mkdir -p $O
timeout 300 python tools/profile_step_host.py cfg2_bunny70k 300 > $O/step_host_70k.txt 2>&1
cat $O/step_host_70k.txt
