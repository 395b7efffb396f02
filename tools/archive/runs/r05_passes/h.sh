#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_h; rm -rf $O; mkdir -p $O/pmc
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o asm -- python $GRAFT_REPO_ROOT/tools/time_assembly.py ) > $O/rocprof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/assembly_kernel_stats.csv && head -16 "$f" | cut -d, -f1-4 | tee $O/kernel_stats_head.txt
rm -rf $O/prof
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/asm_$C -o out -- python $GRAFT_REPO_ROOT/tools/time_assembly.py ) > $O/pmc/asm_$C.log 2>&1
done
python tools/pmc_summary.py $O/pmc/asm_FETCH_SIZE $O/pmc/asm_WRITE_SIZE cfg4_plane1m $O/pmc_assembly.json 2>&1 | tee $O/pmc_summary.txt
find $O/pmc -name "*.csv" -size +1M -delete
