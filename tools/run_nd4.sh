#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for leaf in 48 64 96 128; do for lg in 64 96 160; do for st in 32 64; do
echo -n "leaf $leaf long $lg steps $st: "; LS_ND_LONG=$lg LS_ND_STEPS=$st python tools/nd_prof.py cfg4_plane1m $leaf 50 2>/dev/null | head -1
done; done; done
for leaf in 48 96; do echo -n "dragon leaf $leaf long 96: "; LS_ND_LONG=96 python tools/nd_prof.py cfg3_dragon250k $leaf 50 2>/dev/null | head -1; done
