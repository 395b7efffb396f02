#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_tq; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { ( v=$1; shift; for kv in "$@"; do export "$kv"; done; [ $v != base ] && export LD_LIBRARY_PATH=$PWD/tools/build/v_$v:${LD_LIBRARY_PATH:-}; for n in 1000 1400; do for rep in 1 2; do timeout 200 $D $n $((300000 / n)) 3 -1 2>&1 | grep -E "solve \(mode" | sed -E 's/.*launches +([0-9.]+) us per solve.*/\1/' | tr '\n' ' '; done; echo -n "| "; done; echo " $v $*" ); }
{ run base; run tier_q2; run tier_q8; run base LS_ND_XCD_TIER=0; run base LS_ND_NT=3; run base; } 2>&1 | tee $O/tier_q.txt
