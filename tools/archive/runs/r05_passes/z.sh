#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_z; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { ( export LS_ND_TIER_WAVES=$3; [ "$3" = "" ] && unset LS_ND_TIER_WAVES; timeout 200 $D $1 $((300000 / $1)) 3 $2 2>&1 | grep -E "solve \(mode|hash|error|HIP|tier" | sed 's/max |x.*events/ev/' | tr '\n' ' ' | sed "s/^/[n=$1 tier=$2 waves=${3:-rule}] /"; echo ); }
for rep in 1 2; do
  run 1000 -1 ""; run 1000 4 ""; run 1000 4 16; run 1000 3 16; run 1000 5 16
  run 2000 -1 ""; run 2000 5 16; run 2000 5 ""
  run 500 -1 ""; run 500 3 16; run 500 4 16
  run 1400 -1 ""; run 1400 4 16
done 2>&1 | tee $O/tier16.txt
