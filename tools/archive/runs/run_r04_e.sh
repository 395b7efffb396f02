#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_e; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for lib in default v_rs2048 v_rs1024; do
  for rep in 1 2; do
    if [ $lib = default ]; then timeout 600 python tools/time_dedup.py cfg3_dragon250k cfg4_plane1m cfg2_bunny70k; else LARGESTEPS_HIP_LIB=tools/build/$lib/liblargesteps_hip.so timeout 600 python tools/time_dedup.py cfg3_dragon250k cfg4_plane1m cfg2_bunny70k; fi 2>&1 | grep "^cfg" | sed "s/^/[$lib] /"
  done
done > $O/dedup.txt
cat $O/dedup.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o d -- python $GRAFT_REPO_ROOT/tools/time_dedup.py cfg3_dragon250k ) > $O/rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/dedup_kernel_stats.csv; rm -rf $O/prof
head -12 $O/dedup_kernel_stats.csv | cut -c1-170
