#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/exp; rm -rf $O; mkdir -p $O/pmc
export TMPDIR=/tmp
timeout 300 python tools/bench_normals.py 2>&1 | grep -v amdgpu > $O/normals.txt
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/n_$C -o out -- python tools/bench_normals.py ) > $O/pmc/n_$C.log 2>&1
done
python tools/pmc_summary.py $O/pmc/n_FETCH_SIZE $O/pmc/n_WRITE_SIZE cfg4_plane1m $O/pmc_normals.json > $O/pmc_summary.log 2>&1
( cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o n -- python tools/bench_normals.py ) > $O/rocprof.log 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/normals_kernel_stats.csv; rm -rf $O/prof; find $O/pmc -name "*.csv" -size +1M -delete
cat $O/normals.txt; python - <<'PY'
import json
d=json.load(open("gpurun_out/exp/pmc_normals.json"))
for k,v in d["kernels"].items():
    if k.startswith("ls::") or "ls::" in k: print(k[:60], v["dispatches"], "read MB", round(v["read_bytes_corrected"]/1e6,1), "write MB", round(v["write_bytes"]/1e6,1))
PY
grep "ls::" $O/normals_kernel_stats.csv | cut -d, -f1-4 | head -12
