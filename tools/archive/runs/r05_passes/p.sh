#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_p; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x -k "matrix or assembl or laplacian or geometry or soup or duplicates or abi" 2>&1 | tail -5 | tee $O/pytest_some.txt
bash tools/r05/q.sh
