#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "remove_duplicates or batched" > $O/tests.log 2>&1; tail -5 $O/tests.log
for w in cfg4_plane1m cfg3_dragon250k cfg2_bunny70k; do timeout 600 python tools/bench_remesh.py $w 100 3 2>&1 | grep -v amdgpu.ids; done | tee $O/remesh.txt
timeout 600 python tools/bench_batched.py 64 40 50 2>&1 | grep -v amdgpu.ids | tee $O/batched.txt
timeout 600 python tools/bench_batched.py 256 16 50 2>&1 | grep -v amdgpu.ids | tee -a $O/batched.txt
timeout 600 python tools/bench_step.py cfg4_plane1m 30 2>&1 | grep -v amdgpu.ids | tee $O/step.txt
timeout 600 python tools/bench_step.py cfg3_dragon250k 30 2>&1 | grep -v amdgpu.ids | tee -a $O/step.txt
