#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_knobs; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { ( for kv in "$@"; do export "$kv"; done; for rep in 1 2; do timeout 200 $D 1000 300 3 -1 2>&1 | grep -E "solve \(mode" | sed -E 's/.*launches +([0-9.]+) us per solve.*/\1/' | tr '\n' ' '; done; echo " $*" ); }
{
run BASE=1
run LS_ND_LONG_UP=64; run LS_ND_LONG_UP=128; run LS_ND_LONG_UP=512
run LS_ND_LONG=128; run LS_ND_LONG=32
run LS_ND_BW_LONG=900; run LS_ND_BW_LONG=2100
run LS_ND_TILES=250; run LS_ND_TILES=1000
run LS_ND_STEPS_UP=16; run LS_ND_STEPS_UP=64; run LS_ND_STEPS=64; run LS_ND_STEPS=256
run BASE=1
} 2>&1 | tee $O/knobs_1m_tier16.txt
