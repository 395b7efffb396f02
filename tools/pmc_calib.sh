#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/ubench/pmc_calib.hip), one counter per pass
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/pmc_calib; rm -rf $O; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/$C -o out -- $GRAFT_REPO_ROOT/tools/build/pmc_calib ) > $O/$C.log 2>&1; done
python3 - <<'PY'
import csv, glob
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob(f"gpurun_out/pmc_calib/{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r.get("Counter_Name") == C:
                print(f"{C:10s} {r['Kernel_Name'].split('(')[0]:12s} {float(r['Counter_Value']) * 1024 / 1e6:10.2f} MB (counter value x 1024)")
PY
grep "bytes per kernel" $O/FETCH_SIZE.log
