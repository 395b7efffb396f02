#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_thr; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
t() { timeout 200 $D $1 $((300000 / $1)) 3 $2 2>&1 | grep -E "solve \(mode" | sed -E 's/.*launches +([0-9.]+) us per solve.*/\1/' | tr '\n' ' '; }
{
for n in 800 900 950 1100; do
  echo -n "n=$n  4-wave tier (levels-5): "; for r in 1 2; do LS_ND_TIER_WAVES=4 t $n -1; done
  echo -n " | 16-wave tier (levels-4): "; for r in 1 2; do LS_ND_TIER_WAVES=16 t $n 4; done; echo
done
} 2>&1 | tee $O/threshold.txt
