#!/bin/bash
# round 4, GPU pass G: the factorisation's batched fp64 GEMM with the next slice prefetched into registers, 32- and 16-deep slices
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_g; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for lib in default v_gk16; do
  L=""; [ $lib != default ] && L="LARGESTEPS_HIP_LIB=$PWD/tools/build/$lib/liblargesteps_hip.so"
  ( cd /tmp && env $L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$lib -o ctor -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py cfg4_plane1m ) > $O/rocprof_$lib.log 2>&1
  cp $(find $O/prof_$lib -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats_$lib.csv; rm -rf $O/prof_$lib
  echo "== $lib"; head -6 $O/constructor_kernel_stats_$lib.csv | cut -c1-150
  for w in cfg4_plane1m cfg5_plane4m cfg3_dragon250k cfg2_bunny70k; do env $L timeout 300 python tools/profile_constructor.py $w 4 2>&1 | grep -E "constructor" | sed "s/^/[$lib] /"; done
done 2>&1 | tee $O/summary.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_nested_gpu.py -m gpu -q -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
