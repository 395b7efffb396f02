#!/bin/bash
# round 5, pass C: the product after the laboratory code left direct.hip / nd_tier.h and the cache policy became a per-handle choice:
# solution hashes (must equal round 4's), the rule against forced on / off by size, the GPU suite, the driver-style bench line
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_c; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for rep in 1 2; do
for n in 265 500 700 1000 1400 2000; do
  for nt in rule 0 1; do
    ( [ $nt != rule ] && export ND_DRIVE_NT=$nt; timeout 200 $D $n $((300000 / n)) 3 -1 2>&1 | grep -E "solve \(mode|hash|error|HIP" | sed 's/max |x.*events/ev/' | tr '\n' ' ' | sed "s/^/[nt=$nt n=$n] /"; echo )
  done
done; done 2>&1 | tee $O/nt_rule.txt
timeout 1500 python -m pytest tests/test_nested_gpu.py tests/test_gpu_parity.py -m gpu -q -x -n 3 2>&1 | tail -5 | tee $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 3 2>&1 | grep -v amdgpu | tee $O/bench_cfg4.json | cut -c1-600
