#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_y; rm -rf $O; mkdir -p $O
for w in cfg4_plane1m cfg5_plane4m cfg3_dragon250k cfg2_bunny70k; do timeout 300 python tools/profile_constructor.py $w 6 2>&1 | grep -E "constructor" | tail -4; done > $O/constructor.txt; cat $O/constructor.txt
for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k; do timeout 600 python tools/bench_remesh.py $w 100 6 2>&1 | grep -v amdgpu.ids; done > $O/remesh.txt; cat $O/remesh.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "pool or remesh or close" 2>&1 | tail -3
