#!/bin/bash
# round 5, pass B: which streams carry nt at which size (variant libraries, C-ABI driver); precision study of the factor's storage
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_b; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { # name n solves
  local lib=large-steps-pytorch_amd/lib; [ "$1" != base ] && lib=tools/build/v_$1
  LD_LIBRARY_PATH=$lib timeout 200 $D $2 $3 3 -1 0 2>&1 | grep -E "solve (mode|error|HIP" | sed 's/max |x.*events/ev/' | tr '\n' ' ' | sed "s/^/[$1 n=$2] /"; echo
}
for rep in 1 2; do
for n in 265 500 700 1000 1400 2000; do
  for v in base nt1 nt2 nt3 nt4 nt5 nt6 nt7 nt23 nt31; do run $v $n $((300000 / n)); done
done; done 2>&1 | tee $O/ab_sizes.txt
LARGESTEPS_HIP_LIB=tools/build/liblargesteps_hip_exp.so timeout 1500 python tools/precision_study.py 2>&1 | grep -v amdgpu | tee $O/precision_study.txt
