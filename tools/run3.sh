#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -30 ) > gpurun_out/pytest.log 2>&1
( timeout 500 python tools/sweep.py cfg4_plane1m cfg2_bunny70k ) > gpurun_out/sweep.txt 2>&1
tail -5 gpurun_out/pytest.log
grep -E "block|==|check|warm" gpurun_out/sweep.txt | head -40
