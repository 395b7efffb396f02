#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lg in 100 160 256 400 700 1200; do for st in 32 64 128; do
echo -n "long $lg steps $st: "; LS_ND_LONG=$lg LS_ND_STEPS=$st python tools/nd_prof.py cfg4_plane1m 64 50 4 2>/dev/null | head -1
done; done
