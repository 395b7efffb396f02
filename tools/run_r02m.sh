#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02m; mkdir -p $O
for v in bw8 bw16; do
  export LARGESTEPS_HIP_LIB=$GRAFT_REPO_ROOT/large-steps-pytorch_amd/lib/variants/lib_$v.so
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o nd -- python $GRAFT_REPO_ROOT/tools/nd_prof.py cfg4_plane1m 64 10 ) > $O/prof.log 2>&1
  f=$(find $O/prof -name "*kernel_trace.csv" | head -1); echo "$v: $(python tools/nd_trace.py $f | grep -v tier | awk '{print $6}' | tr '\n' ' ')"
  rm -rf $O/prof
done | tee $O/variants.txt
