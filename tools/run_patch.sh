#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "patch or chebyshev or one_million or determinism" 2>&1 | tail -30 ) > gpurun_out/pytest_patch.log 2>&1
tail -12 gpurun_out/pytest_patch.log
for cfg in "4096,8,6800"; do
  echo "== LARGESTEPS_PATCH=$cfg"
  ( LARGESTEPS_PATCH=$cfg timeout 300 python tools/sweep.py cfg4_plane1m 2>&1 | grep -E "PATCH|one-step" )
done
( timeout 300 python tools/sweep.py cfg5_plane4m 2>&1 | grep -E "PATCH|one-step|==" )
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > /dev/null 2>&1
python - <<'PY'
import csv, collections, statistics
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/pmc3/out_counter_collection.csv')):
    k=r['Kernel_Name'].split('(')[0]
    if 'k_patch' in k: d[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items(): print(k, {c: f"{statistics.median(x):.3g}" for c,x in v.items()})
PY
