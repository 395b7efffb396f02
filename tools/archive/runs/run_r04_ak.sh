#!/bin/bash
# PMC passes: L2 hit rates and the size mix of the read requests that leave the L2, per solve kernel
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ak; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
i=0
for C in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_sum TCC_CYCLE_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p$i -o out -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-baselines ) > $O/p$i.log 2>&1
  grep -i -E "error|invalid|not" $O/p$i.log | head -3
done
python - <<'PY'
import csv, glob, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r04_ak")
out = open(os.path.join(O, "tcc_counters.txt"), "w")
for d in sorted(glob.glob(O + "/p*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "k_nd_" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k in sorted(acc):
            out.write(k + ": " + ", ".join(f"{c} {acc[k][c] / max(1, n[(k, c)]):.4g}" for c in sorted(acc[k])) + "\n")
out.close()
print(open(os.path.join(O, "tcc_counters.txt")).read())
PY
find $O -name "*.csv" -size +200k -delete
