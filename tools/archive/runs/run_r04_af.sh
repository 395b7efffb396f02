#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_af; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for n in 1000 500 250 100 40 2000; do timeout 300 $D $n 30 3 -1 0 2>&1 | grep -E "hash" | sed "s/^/[n=$n] /"; done > $O/hashes.txt
cat $O/hashes.txt
for w in cfg4_plane1m cfg5_plane4m cfg3_dragon250k cfg2_bunny70k; do timeout 300 python tools/profile_constructor.py $w 6 2>&1 | grep -E "constructor" | tail -3; done > $O/constructor.txt; cat $O/constructor.txt
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ctor -o ctor -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py cfg4_plane1m 3 ) > $O/rocprof_ctor.log 2>&1
cp $(find $O/prof_ctor -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats.csv; rm -rf $O/prof_ctor
grep -E "spd|gemm" $O/constructor_kernel_stats.csv | cut -c1-120
timeout 900 python -m pytest tests/test_nested_gpu.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
