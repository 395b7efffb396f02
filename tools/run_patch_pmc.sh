#!/bin/bash
set -u
mkdir -p gpurun_out/pmc_patch
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_patch/a -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_patch/a.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, collections, statistics
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open('gpurun_out/pmc_patch/a/out_counter_collection.csv')):
    k=r['Kernel_Name'].split('(')[0]
    if 'k_patch' in k or 'k_cheb' in k:
        d[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items():
    print(k, {c: f"{statistics.median(x):.3g}" for c,x in v.items()}, 'n', len(next(iter(v.values()))))
PY
tail -3 gpurun_out/pmc_patch/a.log
