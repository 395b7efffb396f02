#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_k; rm -rf $O; mkdir -p $O
timeout 120 tools/build/mall_prefetch > $O/mall_prefetch.txt 2>&1
cat $O/mall_prefetch.txt
