#!/bin/bash
# full GPU pass (round-1 final): parity tests, smoke, bench (default = direct solver; iterative and PCG A/B), rocprofv3
# kernel stats + FETCH_SIZE / WRITE_SIZE passes of the bench command
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/pmc3
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -30 ) > gpurun_out/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 400 python bench.py --steps 50 --warmup 3 ) > gpurun_out/bench.json 2> gpurun_out/bench.err
( timeout 400 python bench.py --steps 20 --warmup 3 --iterative --no-cpu-baseline ) > gpurun_out/bench_iterative.json 2> gpurun_out/bench_iterative.err
( timeout 400 python bench.py --steps 20 --warmup 3 --pcg --no-cpu-baseline ) > gpurun_out/bench_pcg.json 2> gpurun_out/bench_pcg.err
rm -rf gpurun_out/prof_full3
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_full3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline ) > gpurun_out/rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc3/bench_$C
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3/bench_$C -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmc3/bench_$C.log 2>&1
done
python tools/pmc_summary.py gpurun_out/pmc3/bench_FETCH_SIZE gpurun_out/pmc3/bench_WRITE_SIZE cfg4_plane1m gpurun_out/pmc_traffic.json > gpurun_out/pmc_summary.log 2>&1
python tools/nd_trace.py $(find gpurun_out/prof_full3 -name "*kernel_trace.csv" | head -1) > gpurun_out/nd_levels.txt 2>&1
# keep the merge small: the raw per-dispatch csv files are not needed
find gpurun_out/pmc3 -name "*.csv" -size +2M -delete
find gpurun_out/prof_full3 -name "*kernel_trace.csv" -size +8M -delete
tail -4 gpurun_out/pytest.log
tail -2 gpurun_out/smoke.log
cat gpurun_out/bench.json
cut -c1-400 gpurun_out/bench_iterative.json; echo
cut -c1-400 gpurun_out/bench_pcg.json; echo
tail -3 gpurun_out/bench.err
head -8 gpurun_out/prof_full3/bench_kernel_stats.csv | cut -c1-200
cat gpurun_out/pmc_summary.log | head -30
cat gpurun_out/nd_levels.txt
