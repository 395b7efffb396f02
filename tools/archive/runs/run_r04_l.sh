#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_l; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_ctor -o ctor -- python $GRAFT_REPO_ROOT/tools/profile_constructor.py cfg4_plane1m 3 ) > $O/rocprof_ctor.log 2>&1
T=$(find $O/prof_ctor -name "*kernel_trace.csv" | head -1)
python tools/ctor_launches.py $T > $O/ctor_launches.txt 2>&1
python tools/ctor_timeline.py $T > $O/ctor_timeline.txt 2>&1
rm -rf $O/prof_ctor
grep -c . $O/ctor_launches.txt; tail -5 $O/rocprof_ctor.log
