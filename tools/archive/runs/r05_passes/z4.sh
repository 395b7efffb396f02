#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_z4; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
line() { grep -E "^plane|solve \(mode|error|HIP|max" | sed 's/ nnz [0-9]*,//; s/, factor.*down//; s/ev: first/ev/; s/||b - M x.*events://' | tr '\n' ' '; echo; }
for rep in 1 2; do
for n in 500 700 850 1000 1400 2000; do
  ( export ND_DRIVE_PICK=1; timeout 200 $D $n $((300000 / n)) 3 -1 2>&1 | line | cut -c1-300 )
done; done 2>&1 | tee $O/fold.txt
