#!/bin/bash
# round 5, pass F: the assembler with rows sorted in registers and a tile-compact emit: parity (bit-exact fixtures), time, kernel trace
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_f; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compute_matrix or laplacian or soup or duplicates or int32 or golden or surve or one_million" 2>&1 | tail -5 | tee $O/pytest_assembly.txt
timeout 300 python tools/time_assembly.py 2>&1 | grep -v amdgpu | tee $O/time_assembly.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o asm -- python $GRAFT_REPO_ROOT/tools/time_assembly.py > /dev/null 2>&1)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" | cut -d, -f1-8 | tee $O/kernel_stats_head.txt
timeout 1500 python -m pytest tests/ -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_all.txt
