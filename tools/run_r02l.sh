#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
for v in q2 q4 q8; do
  export LARGESTEPS_HIP_LIB=$GRAFT_REPO_ROOT/large-steps-pytorch_amd/lib/variants/lib_$v.so
  for A in 0 4 8; do
  ( cd /tmp && LS_ND_ABLATE=$A timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o nd -- python $GRAFT_REPO_ROOT/tools/nd_prof.py cfg4_plane1m 64 10 ) > $O/prof.log 2>&1
  f=$(find $O/prof -name "*kernel_trace.csv" | head -1); echo "$v ablate=$A $(python tools/nd_trace.py $f | grep "k_nd_tier\|total" | awk '{print $1 $2, $7}' | tr '\n' ' ')"
  rm -rf $O/prof
  done
done | tee $O/variants.txt
