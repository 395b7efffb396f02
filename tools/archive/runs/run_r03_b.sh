#!/bin/bash
# round 3, second GPU pass: the normals pair (tests, times, kernel stats), the constructor with the spinning host pool
# usage (on the GPU box): bash tools/run_r03_b.sh  -> gpurun_out/r03b/
O=gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 python -m pytest tests/test_normals.py "tests/test_gpu_parity.py::test_optimisation_step_as_a_captured_graph" \
    "tests/test_gpu_parity.py::test_optimisation_step_trajectory_vs_reference" -m gpu -x -q > $O/pytest_normals.log 2>&1
tail -5 $O/pytest_normals.log
timeout 300 python tools/bench_normals.py > $O/normals_times.txt 2>&1; grep -v Warn $O/normals_times.txt | tail -4
LARGESTEPS_NORMALS_PAIR=0 timeout 300 python tools/bench_normals.py > $O/normals_times_general.txt 2>&1; grep "^normals" $O/normals_times_general.txt
timeout 300 python tools/bench_step.py > $O/step.txt 2>&1; tail -6 $O/step.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_normals -o n -- python tools/bench_normals.py > /dev/null 2>&1
find $O/prof_normals -name "*kernel_stats.csv" -exec cp {} $O/normals_kernel_stats.csv \;
grep -E "ls::k_" $O/normals_kernel_stats.csv | awk -F'","' '{printf "%-60s %6s %10.1f us\n", substr($1,2,60), $2, $4/1000}'
LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py cfg4_plane1m 3 > $O/constructor.txt 2>&1
grep -E "bisection|push lists|constructor" $O/constructor.txt
nproc
