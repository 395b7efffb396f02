#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_s; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "adam or captured" 2>&1 | tail -4 | tee $O/pytest_adam.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o s -- python $R/tools/bench_step.py cfg4_plane1m 30 ) > $O/step.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/step_kernel_stats.csv; rm -rf $O/prof
grep "per optim" $O/step.log | cut -c1-80
grep adam $O/step_kernel_stats.csv | cut -c1-60,150-260
