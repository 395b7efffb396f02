cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/normals; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_normals -o n -- python $GRAFT_REPO_ROOT/tools/bench_normals.py > $O/prof_normals.log 2>&1
find $O/prof_normals -name "*kernel_stats.csv" -exec cp {} $O/normals_kernel_stats.csv \;
grep -E "ls::k_" $O/normals_kernel_stats.csv | awk -F'","' '{printf "%-90s %6s %10.1f us\n", substr($1,2,90), $2, $4/1000}'
grep "^normals\|stock" $O/prof_normals.log
rm -rf $O/prof_normals
