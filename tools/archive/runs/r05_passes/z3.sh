#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_z3; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
line() { grep -E "^plane|solve \(mode|error|HIP|max" | sed 's/ nnz [0-9]*,//; s/, factor.*down//; s/ev: first.*//' | tr '\n' ' '; echo; }
for n in 850 950 1000 1400 2000; do
  ( export ND_DRIVE_PICK=1; timeout 200 $D $n $((300000 / n)) 3 -1 2>&1 | line | cut -c1-330 )
done 2>&1 | tee $O/rule.txt
timeout 1500 python -m pytest tests/test_nested_gpu.py tests/test_gpu_parity.py -m gpu -q -x -n 3 2>&1 | tail -3 | tee $O/pytest.txt
for i in 1 2 3; do timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-baselines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver-style run', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config'].get('tolerance'))"; done | tee $O/driver_style.txt
