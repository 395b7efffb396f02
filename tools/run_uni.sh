#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 ) > gpurun_out/pytest.log 2>&1
tail -4 gpurun_out/pytest.log
( timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/bench_uni.json 2> gpurun_out/bench_uni.err
cut -c1-1500 gpurun_out/bench_uni.json; tail -3 gpurun_out/bench_uni.err
( LARGESTEPS_EXPLICIT_VALUES=1 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/bench_explicit.json 2> gpurun_out/bench_explicit.err
cut -c1-400 gpurun_out/bench_explicit.json
( timeout 300 python tools/sweep.py cfg2_bunny70k 2>&1 | grep -E "cheby|==" ) > gpurun_out/sweep_uni.txt 2>&1; cat gpurun_out/sweep_uni.txt
