#!/bin/bash
# full GPU pass: parity tests, smoke, bench (default, one-step A/B, PCG A/B), rocprofv3 kernel trace + PMC passes of the bench command
set -u
mkdir -p gpurun_out/pmc2
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -30 ) > gpurun_out/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 400 python bench.py --steps 20 --warmup 3 ) > gpurun_out/bench.json 2> gpurun_out/bench.err
( LARGESTEPS_NO_PATCHES=1 timeout 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline ) > gpurun_out/bench_onestep.json 2> gpurun_out/bench_onestep.err
( timeout 400 python bench.py --steps 20 --warmup 3 --pcg --no-cpu-baseline ) > gpurun_out/bench_pcg.json 2> gpurun_out/bench_pcg.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_full -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline ) > gpurun_out/rocprof.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/bench_$C -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmc2/bench_$C.log 2>&1
done
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2/sq -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/pmc2/sq.log 2>&1
tail -4 gpurun_out/pytest.log
tail -2 gpurun_out/smoke.log
cat gpurun_out/bench.json
cut -c1-300 gpurun_out/bench_onestep.json; echo
cut -c1-300 gpurun_out/bench_pcg.json; echo
tail -3 gpurun_out/bench.err
head -4 gpurun_out/prof_full/bench_kernel_stats.csv | cut -c1-250
