# short pass on the final state of round 6:  gpurun --timeout 1800 -- 'bash tools/archive/runs/r06_passes/final.sh'
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_run2; rm -rf $O; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 ) > $O/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --workload cfg4b_sphere1m ) > $O/bench_driver_style_sphere.json 2> $O/bench_driver_style_sphere.err
( timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 ) > $O/bench_loopback_8_plain_start.json 2> $O/bench_loopback_8_plain_start.err
for w in cfg4_plane1m cfg4b_sphere1m; do timeout 600 python tools/bench_remesh.py $w 100 8 2>&1 | grep -v amdgpu.ids; done > $O/remesh.txt
tail -3 $O/pytest.log; tail -1 $O/smoke.log; cut -c1-260 $O/bench_driver_style.json; echo; cut -c1-260 $O/bench_driver_style_sphere.json; echo; cat $O/remesh.txt
