#!/bin/bash
# the GPU suite with the round's knobs turned the other way: no buffer pool, quarter tiles for every product, three analysis threads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_an; rm -rf $O; mkdir -p $O
( LS_POOL_GB=0 LS_GEMM_SMALL_TILES=100000000 LS_PLAN_THREADS=3 timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_buffer_pool_between_constructions 2>&1 | tail -6 ) > $O/pytest_knobs.log 2>&1
tail -4 $O/pytest_knobs.log
