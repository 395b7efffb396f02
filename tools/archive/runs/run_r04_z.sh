#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "buffer_pool" 2>&1 | tail -3
