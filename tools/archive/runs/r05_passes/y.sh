#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_y; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for rep in 1 2; do
for v in base nd_pipe nd_e4 nd_e1 nd_pipe_e4 nd_pipe_e1; do
  for n in 1000 500 2000; do
    ( [ $v != base ] && export LD_LIBRARY_PATH=$PWD/tools/build/v_$v:${LD_LIBRARY_PATH:-}; timeout 200 $D $n $((300000 / n)) 3 -1 2>&1 | grep -E "solve \(mode|hash|error|HIP" | sed 's/max |x.*events/ev/' | tr '\n' ' ' | sed "s/^/[$v n=$n] /"; echo )
  done
done; done 2>&1 | tee $O/dot_rows_variants.txt
