#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_rep; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-baselines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style run', d['ms_per_step'], d['value'], 'launches', len(d['config']['launches']))"; done | tee $O/driver_style_repeats.txt
