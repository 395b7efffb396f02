#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_x; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_normals.py tests/test_gpu_parity.py -m gpu -q -x -k "normal or captured or optimisation_step or vertex_major" 2>&1 | tail -3 | tee $O/pytest_normals.txt
timeout 300 python tools/bench_normals.py 2>&1 | grep "^normals" | tee $O/normals.txt
timeout 300 python tools/bench_step.py cfg4_plane1m 30 2>&1 | grep "^cfg" | cut -c1-70 | tee $O/step.txt
