#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/ndprof
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ndprof -- python tools/nd_prof.py cfg4_plane1m ${1:-96} 5 2>&1 | grep -v "^W2\|^E2\|rocprofiler" | tail -22
f=$(find gpurun_out/ndprof -name "*kernel_trace.csv" | head -1)
python tools/nd_trace.py "$f"
