#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_u; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_nested_gpu.py tests/test_gpu_parity.py -m gpu -q -x -n 3 2>&1 | tail -3 | tee $O/pytest.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_ctor -o ctor -- python $R/tools/profile_constructor.py cfg4_plane1m ) > $O/rocprof_ctor.log 2>&1
cp $(find $O/prof_ctor -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats.csv; rm -rf $O/prof_ctor
grep "spd_inverse\|gemm_batched" $O/constructor_kernel_stats.csv | cut -c1-40,60-140
grep constructor $O/rocprof_ctor.log | tail -3
