#!/bin/bash
# tier height at 1M and 4M with 4 and 8 waves per tier workgroup
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ab; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for n in 1000 2000; do for t in -1 2 3 4 5; do for W in 4 8; do echo -n "n=$n tier=$t LS_ND_TIER_WAVES=$W: "; LS_ND_TIER_WAVES=$W timeout 300 $D $n 200 3 $t 0 2>&1 | grep -E "persist 0|error" | cut -c1-100; done; done; done > $O/tier.txt 2>&1
cat $O/tier.txt
