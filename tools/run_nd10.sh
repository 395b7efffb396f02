#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "direct or from_differential_vs_reference" 2>&1 | tail -3
for cfgl in "4 64" "4 32" "4 16" "2 32" "8 32"; do set -- $cfgl; python tools/nd_prof.py cfg4_plane1m $2 50 $1 2>/dev/null | head -1; done
echo -n "nopack 4 64: "; LS_ND_NO_PACK=1 python tools/nd_prof.py cfg4_plane1m 64 50 4 2>/dev/null | head -1
python tools/nd_prof.py cfg3_dragon250k 64 50 4 2>/dev/null | head -1
python tools/nd_prof.py cfg3_dragon250k 32 50 4 2>/dev/null | head -1
python tools/nd_prof.py cfg2_bunny70k 32 50 4 2>/dev/null | head -1
mkdir -p gpurun_out/ndprof; rm -rf gpurun_out/ndprof/*
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ndprof -- python tools/nd_prof.py cfg4_plane1m 32 5 4 > /dev/null 2>&1
python tools/nd_trace.py $(find gpurun_out/ndprof -name "*kernel_trace.csv" | head -1)
