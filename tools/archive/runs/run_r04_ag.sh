#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ag; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp && timeout 120 rocprofv3 --list-avail 2>&1 | grep -E "^\s*(Name|name)|TCC_|MALL|DRAM|HBM|EA0" | cut -c1-160 | sort -u | head -120 > $GRAFT_REPO_ROOT/$O/counters.txt
cat $GRAFT_REPO_ROOT/$O/counters.txt | head -100
