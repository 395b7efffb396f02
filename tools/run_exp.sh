#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -8 ) > $O/pytest.log 2>&1
for w in cfg2_bunny70k cfg3_dragon250k; do ( timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines --no-cpu-baseline ) > $O/bench_$w.json 2> $O/bench_$w.err; done
cat $O/pytest.log; for w in cfg2_bunny70k cfg3_dragon250k; do cut -c1-330 $O/bench_$w.json; echo; done
