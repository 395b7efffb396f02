set -x
mkdir -p gpurun_out/r9
O=gpurun_out/r9
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $O/pytest_full.log 2>&1
( timeout 600 python bench.py ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
( timeout 600 python bench.py --workload cfg4b_sphere1m ) > $O/bench_sphere.json 2> $O/bench_sphere.err
LS_PLAN_TIMING=1 timeout 900 python tools/bench_remesh.py cfg4_plane1m 20 10 > $O/remesh_plane.txt 2> $O/remesh_plane_stages.txt
timeout 900 python tools/bench_remesh.py cfg4b_sphere1m 20 6 > $O/remesh_sphere.txt 2>&1
tail -5 $O/pytest_full.log; cut -c1-300 $O/bench_driver_style.json; echo; cut -c1-300 $O/bench_sphere.json; echo; cat $O/remesh_plane.txt $O/remesh_sphere.txt
