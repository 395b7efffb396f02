#!/bin/bash
# scratch: A/B of (braw for the upper levels, fused root)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -5 ) > $O/pytest.log 2>&1
for v in "LS_X=1" "LS_ND_NO_FUSE_ROOT=1" ; do
  for w in cfg4_plane1m cfg2_bunny70k cfg3_dragon250k; do
  echo "== $v $w" >> $O/prof.txt
  ( env $v timeout 300 python tools/nd_prof.py $w 64 300 2>&1 | grep "ms/solve" ) >> $O/prof.txt
  done
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr -o t -- python $GRAFT_REPO_ROOT/tools/nd_prof.py cfg4_plane1m 64 20 ) > $O/rocprof.log 2>&1
python tools/nd_trace.py $(find $O/tr -name "*kernel_trace.csv" | head -1) >> $O/levels.txt 2>&1
rm -rf $O/tr
( timeout 300 python bench.py --steps 50 --warmup 3 --no-cpu-baseline ) > $O/bench.json 2> $O/bench.err
cat $O/pytest.log $O/prof.txt $O/levels.txt; cut -c1-400 $O/bench.json
