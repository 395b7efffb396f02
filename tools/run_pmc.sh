#!/bin/bash
# PMC passes (own runs, kernel-trace only): FETCH_SIZE and WRITE_SIZE for the bench command and for the calibration kernels
set -u
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/bench_$C -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc/bench_$C.log 2>&1
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc/ubench_$C -o out -- python $GRAFT_REPO_ROOT/tools/ubench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/ubench_$C.log 2>&1
done
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/pmc | head -30
for f in gpurun_out/pmc/*/out_counter_collection.csv; do echo $f; head -3 $f | cut -c1-400; done
du -sh gpurun_out/pmc
