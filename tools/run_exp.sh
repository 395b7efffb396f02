#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 300 -k "adam or captured" 2>&1 | tail -15 ) > $O/pytest.log 2>&1
for w in cfg2_bunny70k cfg3_dragon250k cfg4_plane1m; do timeout 300 python tools/bench_step.py $w 50 2>&1 | grep "ms per"; done > $O/step.txt 2>&1
cat $O/pytest.log $O/step.txt
