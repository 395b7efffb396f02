#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_o; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-baselines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style run', d['ms_per_step'], d['value'])"; done | tee $O/driver_style_repeats.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "ordering_argument or laboratory or cache_policy" 2>&1 | tail -3
