#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02k; mkdir -p $O
for S in 0 4 8; do
( cd /tmp && LS_ND_ABLATE=$S timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$S -o nd -- python $GRAFT_REPO_ROOT/tools/nd_prof.py cfg4_plane1m 64 10 ) > $O/prof$S.log 2>&1
f=$(find $O/prof$S -name "*kernel_trace.csv" | head -1); echo "ablate=$S $(python tools/nd_trace.py $f | grep "k_nd_tier\|total" | awk '{print $1 $2, $7}' | tr '\n' ' ')"
rm -rf $O/prof$S
done | tee $O/stagger.txt
