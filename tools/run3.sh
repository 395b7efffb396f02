#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( LS_DEBUG=1 timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 ) > gpurun_out/pytest.log 2>&1
( timeout 500 python tools/sweep.py cfg4_plane1m cfg2_bunny70k cfg3_dragon250k ) > gpurun_out/sweep.txt 2>&1
tail -5 gpurun_out/pytest.log
grep -E "cheby|==|algo 0 block  (256|512) grid  1024" gpurun_out/sweep.txt | head -60
