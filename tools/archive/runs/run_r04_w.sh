#!/bin/bash
# quarter tiles for the small products of the factorisation (default: launches of fewer than 128 tiles of 64 x 64) against LS_GEMM_SMALL_TILES=0 (never) and 512
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_w; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
for n in 1000 500 250 100 40; do timeout 300 $D $n 30 3 -1 0 2>&1 | grep -E "hash" | sed "s/^/[n=$n] /"; done > $O/hashes.txt
cat $O/hashes.txt
for S in 128 0 512 128 0 512; do echo "== LS_GEMM_SMALL_TILES=$S"; LS_GEMM_SMALL_TILES=$S timeout 300 python tools/profile_constructor.py cfg4_plane1m 6 2>&1 | grep -E "constructor" | tail -4; done > $O/constructor.txt
cat $O/constructor.txt
for S in 128 0; do for w in cfg5_plane4m cfg3_dragon250k cfg2_bunny70k; do echo "== LS_GEMM_SMALL_TILES=$S"; LS_GEMM_SMALL_TILES=$S timeout 300 python tools/profile_constructor.py $w 5 2>&1 | grep -E "constructor" | tail -3; done; done > $O/constructor_other.txt; cat $O/constructor_other.txt
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ctor -o ctor -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py cfg4_plane1m 3 ) > $O/rocprof_ctor.log 2>&1
cp $(find $O/prof_ctor -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats.csv; rm -rf $O/prof_ctor
head -6 $O/constructor_kernel_stats.csv | cut -c1-120
