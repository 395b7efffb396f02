#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_z2; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
line() { grep -E "^plane|solve \(mode|error|HIP" | sed 's/max |x.*events/ev/; s/ nnz [0-9]*,//; s/, factor.*down//; s/ev: first.*//' | tr '\n' ' '; echo; }
for n in 130 180 265 350 420 500 550 600 700 850 1000 1200 1400 2000; do
  S=$((300000 / n)); [ $S -gt 2000 ] && S=2000
  echo "== n=$n" 
  ( export ND_DRIVE_PICK=1; timeout 200 $D $n $S 3 -1 2>&1 | line | sed 's/^/   pick          /' )
  L=$( timeout 200 $D $n 1 3 -1 2>&1 | grep "^plane" | sed -E 's/.* ([0-9]+) levels.*/\1/' )
  H=$((L - 4))
  if [ $H -ge 1 ]; then
    ( timeout 200 $D $n $S 3 -1 2>&1 | line | sed 's/^/   arity4 rule   /' )
    ( export LS_ND_TIER_WAVES=16; timeout 200 $D $n $S 3 $H 2>&1 | line | sed "s/^/   arity4 H$H W16 /" )
  fi
done 2>&1 | tee $O/tier16_sizes.txt
