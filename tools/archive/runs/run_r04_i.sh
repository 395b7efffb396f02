#!/bin/bash
# pass I: constructor -- host side (process-wide thread pool, no zero-fill, arenas for the boundary sets), GEMM two slices ahead, two pivots per barrier
# in the in-register inverse (default) against the same build with the one-pivot kernel (v_onepivot: must reproduce the known solution hashes)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_i; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
D=tools/build/nd_drive
run() { lib=$1; n=$2; shift 2; if [ "$lib" = default ]; then env "$@" timeout 300 $D $n 100 3 -1 0; else env LD_LIBRARY_PATH=$PWD/tools/build/$lib "$@" timeout 300 $D $n 100 3 -1 0; fi 2>&1 | grep -E "persist 0|hash|sum of|error|HIP|residual|factor" | sed "s/^/[$lib n=$n] /"; }
( for lib in v_onepivot default; do for n in 1000 500 250 100 40 2000; do run $lib $n X=1; done; done ) > $O/variants.txt 2>&1
cat $O/variants.txt
( timeout 900 python -m pytest tests/test_nested_gpu.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for T in 32 64 128 16; do echo "== LS_PLAN_THREADS=$T"; LS_PLAN_THREADS=$T LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py cfg4_plane1m 5 2>&1 | grep -E "constructor|nd_plan|ls_direct_factor" | tail -22; done > $O/constructor_threads.txt
echo "== one-pivot inverse" >> $O/constructor_threads.txt
LARGESTEPS_HIP_LIB=tools/build/v_onepivot/liblargesteps_hip.so timeout 300 python tools/profile_constructor.py cfg4_plane1m 5 2>&1 | grep -E "constructor" >> $O/constructor_threads.txt
for w in cfg5_plane4m cfg3_dragon250k cfg2_bunny70k; do timeout 300 python tools/profile_constructor.py $w 4 2>&1 | grep -E "constructor"; done > $O/constructor_other.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ctor -o ctor -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py cfg4_plane1m 3 ) > $O/rocprof_ctor.log 2>&1
python tools/ctor_timeline.py $(find $O/prof_ctor -name "*kernel_trace.csv" | head -1) > $O/ctor_timeline.txt 2>&1
cp $(find $O/prof_ctor -name "*kernel_stats.csv" | head -1) $O/constructor_kernel_stats.csv
rm -rf $O/prof_ctor
grep -E "==|constructor" $O/constructor_threads.txt; cat $O/constructor_other.txt; head -12 $O/constructor_kernel_stats.csv | cut -c1-150; tail -32 $O/ctor_timeline.txt
