#!/bin/bash
# PMC passes (counters alone with the kernel trace): LDS conflicts and instruction mix of the solve's kernels
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ai; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p$i -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-baselines ) > $O/p$i.log 2>&1
  tail -2 $O/p$i.log | cut -c1-200
done
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04_ai")
out = open(os.path.join(O, "sq_counters.txt"), "w")
for d in sorted(glob.glob(O + "/p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_nd_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in sorted(acc):
            out.write(k + ": " + ", ".join(f"{c} {acc[k][c] / max(1, n[(k, c)]):.3g} per launch" for c in sorted(acc[k])) + "\n")
out.close()
print(open(os.path.join(O, "sq_counters.txt")).read())
PY
find $O -name "*.csv" -size +200k -delete
