#!/bin/bash
# round 5, pass K: trial cuts (ND_ORDER_MINSEP) on the device against the host rounds; constructor times of the folded configs
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_k; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_nested_gpu.py -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest_nested.txt
for w in scroll250k folded250k cfg3_dragon250k; do
  LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py $w 3 2>&1 | grep -E "constructor|suspect|bisection|positions" | tail -12
  LS_ND_HOST_TRIALS=1 timeout 300 python tools/profile_constructor.py $w 3 2>&1 | grep -E "constructor" | sed 's/^/[host trials] /'
done 2>&1 | tee $O/constructor.txt
for w in cfg3_dragon250k cfg2_bunny70k; do
  LS_ND_ORDER=1 timeout 300 python tools/profile_constructor.py $w 3 2>&1 | grep -E "constructor" | sed 's/^/[LS_ND_ORDER=1 device] /'
  LS_ND_ORDER=1 LS_ND_HOST_TRIALS=1 timeout 300 python tools/profile_constructor.py $w 3 2>&1 | grep -E "constructor" | sed 's/^/[LS_ND_ORDER=1 host] /'
done 2>&1 | tee -a $O/constructor.txt
