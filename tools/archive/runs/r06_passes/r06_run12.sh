set -x
mkdir -p gpurun_out/r12
O=gpurun_out/r12
python -m pytest tests/test_nested_gpu.py tests/test_gpu_parity.py -m gpu -x -q -k "nested or plan or ordering or folded or scroll or shells or misleading or remesh or trial or graph" > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log
LS_PLAN_TIMING=1 python tools/ctor_in_loop.py cfg4b_sphere1m 5 > $O/ctor_in_loop_sphere_stages.txt 2>&1
python tools/ctor_in_loop.py scroll1m 5 > $O/ctor_in_loop_scroll.txt 2>&1
tail -4 $O/pytest.log; grep -v "^\[" $O/ctor_in_loop_sphere_stages.txt | tail -8; tail -8 $O/ctor_in_loop_scroll.txt; awk '/\[nd_plan\] positions/{c++} c==7' $O/ctor_in_loop_sphere_stages.txt | head -8
