#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "direct or from_differential_vs_reference" 2>&1 | tail -3
for kb in 1 16 24 40 64; do echo -n "small_kb $kb: "; LS_ND_SMALL_KB=$kb python tools/nd_prof.py cfg4_plane1m 64 50 4 2>/dev/null | head -1; done
echo -n "leaf32 kb40: "; python tools/nd_prof.py cfg4_plane1m 32 50 4 2>/dev/null | head -1
echo -n "leaf32 kb40 arity2: "; python tools/nd_prof.py cfg4_plane1m 32 50 2 2>/dev/null | head -1
python tools/nd_prof.py cfg3_dragon250k 64 50 4 2>/dev/null | head -1
python tools/nd_prof.py cfg2_bunny70k 64 50 4 2>/dev/null | head -1
mkdir -p gpurun_out/ndprof; rm -rf gpurun_out/ndprof/*
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ndprof -- python tools/nd_prof.py cfg4_plane1m 64 5 4 > /dev/null 2>&1
python tools/nd_trace.py $(find gpurun_out/ndprof -name "*kernel_trace.csv" | head -1)
