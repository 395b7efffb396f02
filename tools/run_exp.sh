#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 2>&1 | tail -5 ) > $O/pytest.log 2>&1
for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k; do LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py $w 2>&1 | grep -E "constructor|nd_plan\] [a-z]"; done > $O/constructor_times.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ctor -o ctor -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py cfg4_plane1m ) > $O/rocprof_ctor.log 2>&1
cp $(find $O/prof_ctor -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats.csv; rm -rf $O/prof_ctor
cat $O/pytest.log $O/constructor_times.txt; head -8 $O/constructor_kernel_stats.csv | cut -c1-140
