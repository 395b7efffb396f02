set -x
mkdir -p gpurun_out/r7
R=$GRAFT_REPO_ROOT; O=gpurun_out/r7
export TMPDIR=/tmp
python tools/irregular_1m.py 300 --quick > $O/quick.txt 2>&1
for w in cfg4_plane1m cfg4b_sphere1m; do
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$w -o bench -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra-baselines --workload $w ) > $O/rocprof_$w.log 2>&1
python tools/nd_trace.py $(find $O/prof_$w -name "*kernel_trace.csv" | head -1) > $O/nd_levels_$w.txt 2>&1
cp $(find $O/prof_$w -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$w.csv
rm -rf $O/prof_$w
done
LARGESTEPS_HIP_LIB=$PWD/tools/build/v_stamps/liblargesteps_hip.so python tools/tier_stamps.py cfg4_plane1m > $O/stamps_plane16.txt 2>&1
cat $O/quick.txt $O/nd_levels_*.txt; grep -v "leaf [0-9] done" $O/stamps_plane16.txt
