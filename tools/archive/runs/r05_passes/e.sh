#!/bin/bash
# round 5, pass E: 24-bit storage of the dense factor blocks (FMODE 2) against fp32 -- error and time by size, C-ABI driver
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_e; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { # label n solves env...
  local label=$1 n=$2 sol=$3; shift 3
  env "$@" timeout 200 $D $n $sol 3 -1 2>&1 | grep -E "solve \(mode|hash|error|HIP|factor " | sed 's/events.*//' | tr '\n' ' ' | sed "s/^/[$label n=$n] /"; echo
}
for rep in 1 2; do
for n in 150 300 500 700 1000 1400 2000; do
  run fp32 $n $((300000 / n)) LS_ND_FACTOR_BITS=32
  run f24 $n $((300000 / n)) LS_ND_FACTOR_BITS=24
  run rule $n $((300000 / n)) X=1
done; done 2>&1 | tee $O/f24_sizes.txt
for n in 1000 2000; do
  run f24_longup64 $n $((300000 / n)) LS_ND_FACTOR_BITS=24 LS_ND_LONG_UP=64
  run f24_longup64 $n $((300000 / n)) LS_ND_FACTOR_BITS=24 LS_ND_LONG_UP=64
done 2>&1 | tee $O/f24_knobs.txt
ND_DRIVE_TABLE=1 LS_ND_FACTOR_BITS=24 timeout 200 $D 1000 200 3 -1 2>&1 | grep -E "levels|sum of" | tee $O/table_f24_1m.txt
timeout 900 python -m pytest tests/test_nested_gpu.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
