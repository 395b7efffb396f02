#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "direct or kernel_shapes or widths or cholesky or configs_vs_oracle or cfg1 or determinism" > $O/tests.log 2>&1; tail -3 $O/tests.log
for c in cfg4_plane1m cfg2_bunny70k cfg3_dragon250k; do timeout 300 python tools/nd_prof.py $c 64 100 2>&1 | grep -E "ms/solve"; done | tee $O/times.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o nd -- python $GRAFT_REPO_ROOT/tools/nd_prof.py cfg4_plane1m 64 20 ) > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/nd_trace.py $f > $O/levels.txt 2>&1; cat $O/levels.txt
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
