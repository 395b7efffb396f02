#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( LS_DEBUG=1 timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 ) > gpurun_out/pytest.log 2>&1
tail -4 gpurun_out/pytest.log; grep -c "stale HIP" gpurun_out/pytest.log
( timeout 500 python tools/sweep.py cfg4_plane1m cfg2_bunny70k cfg1_icosphere2k 2>&1 | grep -E "graph|==|chebyshev block  512 grid  1024|warm" ) > gpurun_out/sweep_graph.txt 2>&1; cat gpurun_out/sweep_graph.txt
( timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err
cut -c1-330 gpurun_out/bench_graph.json; tail -2 gpurun_out/bench_graph.err
