set -x
mkdir -p gpurun_out/r8
( timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r8/pytest_full.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r8/smoke.log 2>&1
( timeout 600 python bench.py ) > gpurun_out/r8/bench_driver_style.json 2> gpurun_out/r8/bench_driver_style.err
tail -5 gpurun_out/r8/pytest_full.log; tail -2 gpurun_out/r8/smoke.log; cut -c1-300 gpurun_out/r8/bench_driver_style.json
