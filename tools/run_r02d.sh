#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "direct or kernel_shapes or widths or cholesky or configs_vs_oracle or cfg1 or determinism" > $O/tests.log 2>&1; tail -3 $O/tests.log
for A in 0 4 8; do
( cd /tmp && LS_ND_ABLATE=$A timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$A -o nd -- python $GRAFT_REPO_ROOT/tools/nd_prof.py cfg4_plane1m 64 10 ) > $O/prof$A.log 2>&1
f=$(find $O/prof$A -name "*kernel_trace.csv" | head -1); echo "ablate=$A $(python tools/nd_trace.py $f | grep "k_nd_tier\|total" | awk '{print $1 $2, $7}' | tr '\n' ' ')"
rm -rf $O/prof$A
done | tee $O/ablate.txt
