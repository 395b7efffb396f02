#!/bin/bash
# kernel times of the solver's constructor (rocprofv3 --kernel-trace --stats of tools/profile_constructor.py) -> gpurun_out/r03b/ctor_kernel_stats.csv
O=$GRAFT_REPO_ROOT/gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ctor -o c -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py ${1:-cfg4_plane1m} > $O/prof_ctor.log 2>&1
find $O/prof_ctor -name "*kernel_stats.csv" -exec cp {} $O/ctor_kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.reader(open("$O/ctor_kernel_stats.csv")))
for r in rows[1:14]:
    print(f"{r[0][:64]:64s} n={r[1]:>5s} total {float(r[2])/1e6:8.2f} ms avg {float(r[3])/1000:9.1f} us max {float(r[6])/1000:9.1f}")
PY
grep constructor $O/prof_ctor.log
rm -rf $O/prof_ctor
