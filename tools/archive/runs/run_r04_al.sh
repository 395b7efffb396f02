#!/bin/bash
# from which reduction length on does the up sweep use the lanes-along-the-reduction kernel (now without LDS bank conflicts)? LS_ND_LONG_UP sweep; same for the down sweep (LS_ND_LONG)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_al; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for rep in 1 2 3; do for L in 256 200 100 64; do echo -n "LS_ND_LONG_UP=$L n=1000: "; LS_ND_LONG_UP=$L timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "persist 0" | cut -c1-70; done; done > $O/long_up.txt 2>&1
for L in 256 200 100 64; do for n in 2000 700 500; do echo -n "LS_ND_LONG_UP=$L n=$n: "; LS_ND_LONG_UP=$L timeout 300 $D $n 200 3 -1 0 2>&1 | grep -E "persist 0" | cut -c1-70; done; done >> $O/long_up.txt 2>&1
cat $O/long_up.txt
