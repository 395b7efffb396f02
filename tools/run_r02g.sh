#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "direct or kernel_shapes or widths or cholesky or configs_vs_oracle or cfg1 or determinism or foreign" > $O/tests.log 2>&1; tail -5 $O/tests.log
for c in cfg4_plane1m cfg2_bunny70k cfg3_dragon250k cfg5_plane4m; do timeout 300 python tools/nd_prof.py $c 64 50 2>&1 | grep -E "ms/solve|constructor"; done | tee $O/times.txt
LS_ND_PYTHON_FACTOR=1 timeout 300 python tools/nd_prof.py cfg4_plane1m 64 50 2>&1 | grep -E "ms/solve|constructor" | sed 's/^/python factor: /' | tee -a $O/times.txt
