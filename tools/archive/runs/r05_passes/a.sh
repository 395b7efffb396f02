#!/bin/bash
# round 5, pass A: cache policy (nt) of the read-once factor streams and 16-byte-aligned rows of the upper levels -- A/B of variant
# libraries through the C-ABI driver (tools/nd_drive.cpp). Solution hashes must agree bit for bit.
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_a; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { # name n solves
  local lib=large-steps-pytorch_amd/lib; [ "$1" != base ] && lib=tools/build/v_$1
  LD_LIBRARY_PATH=$lib timeout 200 $D $2 $3 3 -1 0 2>&1 | grep -E "solve (mode|hash|error|HIP" | tr '\n' ' ' | sed "s/^/[$1 n=$2] /"; echo
}
for rep in 1 2; do
  for v in base nt1 nt3 nt7 nt12 nt28 nt31 pad4 pad4nt31; do run $v 1000 300; done
done 2>&1 | tee $O/ab_1m.txt
for v in base nt3 nt31 pad4 pad4nt31; do run $v 2000 100; run $v 500 500; done 2>&1 | tee $O/ab_4m_250k.txt
export ND_DRIVE_TABLE=1
for v in base nt31 pad4nt31; do
  lib=large-steps-pytorch_amd/lib; [ "$v" != base ] && lib=tools/build/v_$v
  echo "=== $v"; LD_LIBRARY_PATH=$lib timeout 200 $D 1000 200 3 -1 0 2>&1 | grep -E "levels|sum of"
done 2>&1 | tee $O/tables_1m.txt
