#!/bin/bash
# full GPU pass: parity tests, smoke, bench (default + PCG A/B), rocprofv3 kernel trace of the bench command
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 ) > gpurun_out/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 400 python bench.py --steps 20 --warmup 3 ) > gpurun_out/bench.json 2> gpurun_out/bench.err
( timeout 400 python bench.py --steps 20 --warmup 3 --pcg --no-cpu-baseline ) > gpurun_out/bench_pcg.json 2> gpurun_out/bench_pcg.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_full -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/rocprof.log 2>&1
tail -4 gpurun_out/pytest.log
tail -2 gpurun_out/smoke.log
cat gpurun_out/bench.json
cat gpurun_out/bench_pcg.json | cut -c1-900
tail -3 gpurun_out/bench.err gpurun_out/bench_pcg.err
head -6 gpurun_out/prof_full/bench_kernel_stats.csv | cut -c1-220
