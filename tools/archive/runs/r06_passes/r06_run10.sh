set -x
mkdir -p gpurun_out/r10
O=gpurun_out/r10
python tools/ctor_in_loop.py cfg4_plane1m 8 > $O/ctor_in_loop_plane.txt 2>&1
LS_PLAN_TIMING=1 python tools/ctor_in_loop.py cfg4_plane1m 4 > $O/ctor_in_loop_plane_stages.txt 2>&1
LS_PLAN_TIMING=1 python tools/ctor_in_loop.py cfg4b_sphere1m 4 > $O/ctor_in_loop_sphere_stages.txt 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "trial_cuts_are_chosen or laboratory" > $O/pytest.log 2>&1
python tools/bench_step.py cfg2_bunny70k 50 > $O/step.txt 2>&1; python tools/bench_step.py cfg3_dragon250k 50 >> $O/step.txt 2>&1
cat $O/ctor_in_loop_plane.txt; grep -v "^\[" $O/ctor_in_loop_plane_stages.txt | tail -20; tail -3 $O/pytest.log; grep "^cfg" $O/step.txt
