#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ao; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for n in 1000 250 40; do timeout 300 $D $n 30 3 -1 0 2>&1 | grep -E "hash" | sed "s/^/[n=$n] /"; done | tee $O/hashes.txt
timeout 600 python tools/time_dedup.py cfg2_bunny70k cfg3_dragon250k cfg4_plane1m 2>&1 | grep -v amdgpu | tee $O/dedup.txt
timeout 900 python -m pytest tests/test_nested_gpu.py  tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
