#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_am; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o rm -- python $GRAFT_REPO_ROOT/tools/time_dedup.py cfg4_plane1m ) > $O/rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/prof
head -25 $O/kernel_stats.csv | cut -c1-150
tail -5 $O/rocprof.log | cut -c1-200
