#!/bin/bash
# gpurun call 2: full GPU test suite, bench, knob sweep, rocprofv3 kernel trace (csv)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -60 ) > gpurun_out/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 400 python bench.py --steps 20 --warmup 3 ) > gpurun_out/bench.json 2> gpurun_out/bench.err
( timeout 500 python tools/sweep.py ) > gpurun_out/sweep.txt 2>&1
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/rocprof.log 2>&1
ls -R gpurun_out/prof2 | head
tail -25 gpurun_out/pytest.log
tail -3 gpurun_out/smoke.log
cat gpurun_out/bench.json
grep -E "block|==" gpurun_out/sweep.txt | head -40
