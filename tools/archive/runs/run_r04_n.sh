#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_n; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for n in 1000 500 250 100 40 2000; do timeout 300 $D $n 50 3 -1 0 2>&1 | grep -E "hash|persist 0" | sed "s/^/[n=$n] /"; done > $O/hashes.txt
cat $O/hashes.txt
LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py cfg4_plane1m 6 2>&1 | grep -E "constructor|ls_direct_create|numeric" | tail -12 > $O/constructor.txt
cat $O/constructor.txt
for w in cfg5_plane4m cfg3_dragon250k cfg2_bunny70k; do timeout 300 python tools/profile_constructor.py $w 4 2>&1 | grep -E "constructor"; done > $O/constructor_other.txt; cat $O/constructor_other.txt
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
