#!/bin/bash
# the short pass on the round's final state (after run 3 only the normals' forward changed): suite, smoke, the driver's line, step and normals timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_run4; rm -rf $O; mkdir -p $O
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > $O/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 600 python bench.py ) > $O/bench_driver_style.json 2> $O/bench_driver_style.err
( timeout 600 python bench.py --steps 50 --warmup 3 --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
( timeout 600 python tools/bench_step.py cfg4_plane1m 30; timeout 600 python tools/bench_step.py cfg3_dragon250k 30; timeout 600 python tools/bench_step.py cfg2_bunny70k 30 ) 2>&1 | grep "^cfg" > $O/step.txt
timeout 300 python tools/bench_normals.py 2>&1 | grep "^normals\|^stock\|^max" > $O/normals.txt
tail -3 $O/pytest.log; tail -1 $O/smoke.log; cut -c1-300 $O/bench_driver_style.json; echo; cut -c1-200 $O/bench.json; echo; cat $O/step.txt $O/normals.txt
