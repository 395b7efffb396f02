#!/bin/bash
# round 4, first GPU pass: parity on the new state (folded surfaces, zero-spill kernels, experiments library), headline bench line,
# the folded-surface workloads, tier stamps of the unchanged tier kernels (baseline for the LDS-DMA variants)
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_a; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 ) > $O/pytest.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 600 python bench.py --steps 50 --warmup 3 ) > $O/bench.json 2> $O/bench.err
for w in scroll250k scroll10_250k folded250k shells250k cfg3_dragon250k; do ( timeout 400 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines --no-cpu-baseline ) > $O/bench_$w.json 2> $O/bench_$w.err; done
( LS_ND_ORDER=1 timeout 400 python bench.py --steps 50 --warmup 3 --workload cfg3_dragon250k --no-extra-baselines --no-cpu-baseline ) > $O/bench_cfg3_trialcuts.json 2> $O/bench_cfg3_trialcuts.err
( LS_ND_ORDER=1 timeout 400 python bench.py --steps 50 --warmup 3 --workload cfg2_bunny70k --no-extra-baselines --no-cpu-baseline ) > $O/bench_cfg2_trialcuts.json 2> $O/bench_cfg2_trialcuts.err
( timeout 400 python bench.py --steps 50 --warmup 3 --workload cfg2_bunny70k --no-extra-baselines --no-cpu-baseline ) > $O/bench_cfg2_bunny70k.json 2> $O/bench_cfg2.err
for w in scroll250k cfg4_plane1m; do LS_PLAN_TIMING=1 timeout 300 python tools/profile_constructor.py $w 3 2>&1 | grep -E "constructor|nd_plan|ls_direct_factor"; done > $O/constructor_times.txt
( LARGESTEPS_HIP_LIB=tools/build/liblargesteps_hip_exp.so timeout 300 python tools/tier_stamps.py ) > $O/tier_stamps_baseline.txt 2>&1
tail -4 $O/pytest.log; tail -2 $O/smoke.log; cut -c1-400 $O/bench.json; echo
for f in $O/bench_*.json; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    c = d["config"]
    print("  ms", round(d["ms_per_step"], 4), "method", c.get("method"), "dissection", c.get("dissection"), "factor_s", c.get("factor_seconds"), "err", c.get("max_abs_err_vs_v"))
except Exception as e:
    print("  failed:", e)
PY
done
grep -E "constructor" $O/constructor_times.txt | head; cat $O/tier_stamps_baseline.txt | tail -40
