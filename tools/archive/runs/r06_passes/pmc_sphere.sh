# PMC traffic of the solve on the 1M closed noisy sphere:  gpurun --timeout 900 -- 'bash tools/archive/runs/r06_passes/pmc_sphere.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/pmc_sphere; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$O/bench_$C -o out -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-baselines --workload cfg4b_sphere1m ) > $O/bench_$C.log 2>&1
done
python tools/pmc_summary.py $O/bench_FETCH_SIZE $O/bench_WRITE_SIZE cfg4b_sphere1m $O/pmc_traffic_sphere1m.json > $O/pmc_summary.log 2>&1
find $O -name "*.csv" -size +1M -delete
grep "k_nd" $O/pmc_summary.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-baselines --workload cfg4b_sphere1m 2>/dev/null > $O/bench_sphere.json
