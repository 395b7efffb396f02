#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_r; rm -rf $O; mkdir -p $O
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o s -- python $R/tools/bench_step.py cfg4_plane1m 30 ) > $O/step.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/step_kernel_stats.csv; rm -rf $O/prof
grep "per optim" $O/step.log
