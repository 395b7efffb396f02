#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_t; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x -k "normal or captured" 2>&1 | tail -3 | tee $O/pytest_normals.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o n -- python $R/tools/bench_normals.py ) > $O/normals.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/normals_kernel_stats.csv; rm -rf $O/prof
grep "^normals\|max" $O/normals.log
python - $O/normals_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ls::' in r['Name'] and int(r['Calls']) > 20: print("   ", r['Name'].replace('void ls::','')[:40], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
