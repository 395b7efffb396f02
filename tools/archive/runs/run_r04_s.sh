#!/bin/bash
# timing experiments (experiments library): LS_ND_ABLATE bits of the down-sweep level kernel, per-launch table of the C-ABI driver
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_s2; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive_exp
for ab in 0 16 0 16 0 16; do echo "== LS_ND_ABLATE=$ab"; LS_ND_ABLATE=$ab timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "persist 0"; done > $O/alias.txt 2>&1
for ab in 0 16; do echo "== LS_ND_ABLATE=$ab"; LS_ND_ABLATE=$ab ND_DRIVE_TABLE=1 timeout 300 $D 1000 300 3 -1 0 2>&1 | grep -E "levels [0-9]"; done >> $O/alias.txt 2>&1
cat $O/alias.txt
