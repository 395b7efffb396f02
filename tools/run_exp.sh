#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -4 ) > $O/pytest.log 2>&1
for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k; do timeout 300 python tools/profile_constructor.py $w 4 2>&1 | grep -E "constructor"; done > $O/constructor_times.txt
LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py cfg4_plane1m 2 2>&1 | grep -E "ls_direct_factor|nd_plan\] [a-z]" | tail -16 >> $O/constructor_times.txt
cat $O/pytest.log $O/constructor_times.txt
