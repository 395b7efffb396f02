#!/bin/bash
# round 5, pass D: precision study, dense blocks only; to_differential's kernel; the pool test alone
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_d; rm -rf $O; mkdir -p $O
PS_ONLY="fp32,tier+upper,upper,tier dense" LARGESTEPS_HIP_LIB=tools/build/liblargesteps_hip_exp.so timeout 1500 python tools/precision_study.py 2>&1 | grep -v amdgpu | tee $O/precision_study_dense.txt
timeout 600 python tools/time_spmv.py 2>&1 | grep -v amdgpu | tee $O/spmv.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "buffer_pool or cache_policy or laboratory" 2>&1 | tail -30 | tee $O/pytest_some.txt
