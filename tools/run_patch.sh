#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "patch or chebyshev or one_million or determinism" 2>&1 | tail -30 ) > gpurun_out/pytest_patch.log 2>&1
tail -12 gpurun_out/pytest_patch.log
for cfg in "4096,8,6800" "2048,6,3400" "2048,8,6800"; do
  echo "== LARGESTEPS_PATCH=$cfg"
  ( LARGESTEPS_PATCH=$cfg timeout 300 python tools/sweep.py cfg4_plane1m 2>&1 | grep -E "PATCH|one-step" )
done
( timeout 300 python tools/sweep.py cfg2_bunny70k cfg5_plane4m 2>&1 | grep -E "PATCH|one-step|==" )
