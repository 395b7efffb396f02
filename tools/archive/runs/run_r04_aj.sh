#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_aj; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p1 -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-baselines ) > $O/p1.log 2>&1
for w in cfg3_dragon250k; do ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p2 -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-baselines --workload $w ) > $O/p2.log 2>&1; done
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04_aj")
out = open(os.path.join(O, "sq_counters.txt"), "w")
for d in sorted(glob.glob(O + "/p*")):
    if not os.path.isdir(d): continue
    out.write("== " + os.path.basename(d) + (" (1M plane)" if d.endswith("p1") else " (250k cot config, arity 8)") + "\n")
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "ls::" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in sorted(acc):
            if acc[k].get("SQ_LDS_IDX_ACTIVE", 0) <= 0: continue
            out.write(k + ": " + ", ".join(f"{c} {acc[k][c] / max(1, n[(k, c)]):.3g}" for c in sorted(acc[k])) + f"  -> conflicts / active {acc[k]['SQ_LDS_BANK_CONFLICT'] / acc[k]['SQ_LDS_IDX_ACTIVE']:.2f}\n")
out.close()
print(open(os.path.join(O, "sq_counters.txt")).read())
PY
find $O -name "*.csv" -size +200k -delete
