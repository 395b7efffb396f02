#!/bin/bash
# the up sweep of the top levels with the row-per-lane kernel (reads the array the down sweep reads) instead of the lanes-along-the-reduction one
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_t; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for rep in 1 2 3; do for L in 256 600 100000; do echo -n "LS_ND_LONG_UP=$L n=1000: "; LS_ND_LONG_UP=$L timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "persist 0" | cut -c1-120; done; done > $O/long_up.txt 2>&1
for L in 256 100000; do echo "== LS_ND_LONG_UP=$L"; LS_ND_LONG_UP=$L ND_DRIVE_TABLE=1 timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "levels [0-9]|hash"; done >> $O/long_up.txt 2>&1
for n in 2000 500 250; do for L in default 100000; do echo -n "LS_ND_LONG_UP=$L n=$n: "; if [ $L = default ]; then timeout 300 $D $n 200 3 -1 0; else LS_ND_LONG_UP=$L timeout 300 $D $n 200 3 -1 0; fi 2>&1 | grep -E "persist 0" | cut -c1-120; done; done >> $O/long_up.txt 2>&1
cat $O/long_up.txt
