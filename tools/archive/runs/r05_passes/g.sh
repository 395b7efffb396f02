#!/bin/bash
# round 5, pass G: level-kernel shape knobs at 1M / 4M on top of the non-temporal factor streams (C-ABI driver, product library)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_g; rm -rf $O; mkdir -p $O
D=tools/build/nd_drive
run() { local label=$1 n=$2 sol=$3; shift 3
  env "$@" timeout 200 $D $n $sol 3 -1 2>&1 | grep -E "solve \(mode" | sed 's/max |x.*//' | tr '\n' ' ' | sed "s/^/[$label n=$n] /"; echo; }
for rep in 1 2 3; do
  run default 1000 300 X=1
  run long_up64 1000 300 LS_ND_LONG_UP=64
  run long_up128 1000 300 LS_ND_LONG_UP=128
  run bw_long900 1000 300 LS_ND_BW_LONG=900
  run bw_long2100 1000 300 LS_ND_BW_LONG=2100
  run tiles250 1000 300 LS_ND_TILES=250
  run tiles1000 1000 300 LS_ND_TILES=1000
  run steps_up64 1000 300 LS_ND_STEPS_UP=64
  run steps_up16 1000 300 LS_ND_STEPS_UP=16
  run bw900_tiles250 1000 300 LS_ND_BW_LONG=900 LS_ND_TILES=250
  run long_up64_bw900 1000 300 LS_ND_LONG_UP=64 LS_ND_BW_LONG=900
done 2>&1 | tee $O/knobs_1m.txt
for rep in 1 2; do
  run default 2000 100 X=1
  run long_up64 2000 100 LS_ND_LONG_UP=64
  run bw_long900 2000 100 LS_ND_BW_LONG=900
  run tiles1000 2000 100 LS_ND_TILES=1000
  run default 500 500 X=1
  run long_up64 500 500 LS_ND_LONG_UP=64
done 2>&1 | tee $O/knobs_other.txt
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "compute_matrix or laplacian or soup or duplicates or int32 or golden or surve or one_million" 2>&1 | tail -3 | tee $O/pytest_assembly.txt
timeout 300 python tools/time_assembly.py 2>&1 | grep -v amdgpu | tee $O/time_assembly.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o asm -- python $GRAFT_REPO_ROOT/tools/time_assembly.py ) > $O/rocprof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/assembly_kernel_stats.csv && head -16 "$f" | cut -d, -f1-4 | tee $O/kernel_stats_head.txt
rm -rf $O/prof
