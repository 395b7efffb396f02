#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "direct or nested or cfg or tier or kernel_shapes or batched" 2>&1 | tail -4 ) > $O/pytest.log 2>&1
for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k; do timeout 300 python tools/nd_prof.py $w 64 300 2>&1 | grep "ms/solve"; done > $O/prof.txt
timeout 300 python tools/tier_stamps.py cfg4_plane1m 2>&1 | grep -E "phase|leaf 3" > $O/stamps.txt
cat $O/pytest.log $O/prof.txt $O/stamps.txt
