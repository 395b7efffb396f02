#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_w; rm -rf $O; mkdir -p $O
for v in base nrm_f4b3 nrm_f8b4 nrm_f3b2 nrm_f6b6; do
  if [ $v = base ]; then unset LARGESTEPS_HIP_LIB; else export LARGESTEPS_HIP_LIB=$R/tools/build/v_$v/liblargesteps_hip.so; fi
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o n -- python $R/tools/bench_normals.py ) > $O/normals_$v.log 2>&1
  echo "== $v" | tee -a $O/kernels.txt
  grep "^normals" $O/normals_$v.log | tee -a $O/kernels.txt
  python - $(find $O/prof_$v -name "*kernel_stats.csv" | head -1) <<'PY' | tee -a $O/kernels.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'gather' in r['Name'] and 'ls::' in r['Name'] and int(r['Calls']) > 20: print("   ", r['Name'].replace('void ls::','')[:44], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
  rm -rf $O/prof_$v
done
