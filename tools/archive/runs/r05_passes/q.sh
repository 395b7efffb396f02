#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05_q; rm -rf $O; mkdir -p $O
for v in base; do
  if [ $v = base ]; then unset LARGESTEPS_HIP_LIB; else export LARGESTEPS_HIP_LIB=$R/tools/build/v_$v/liblargesteps_hip.so; fi
  echo "== $v" | tee -a $O/variants.txt
  timeout 120 python tools/time_assembly.py 2>&1 | grep compute_matrix | tee -a $O/variants.txt
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o asm -- python $R/tools/time_assembly.py ) > /dev/null 2>&1
  python - $(find $O/prof_$v -name "*kernel_stats.csv" | head -1) <<'PY' | tee -a $O/variants.txt
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print("   ", r['Name'].replace('void ls::','')[:34], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
  rm -rf $O/prof_$v
done
