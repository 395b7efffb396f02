#!/bin/bash
# round 5, pass L: trial cuts as the automatic choice between 12k and 300k vertices: GPU suite, bench lines of cfg2 / cfg3, remesh cycle
set -u
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05_l; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest.txt
for w in cfg3_dragon250k cfg2_bunny70k scroll250k; do
  timeout 300 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines > $O/bench_$w.json 2> $O/err_$w.txt
  LS_ND_ORDER=0 timeout 300 python bench.py --steps 50 --warmup 3 --workload $w --no-extra-baselines --no-cpu-baseline > $O/bench_${w}_longest.json 2>> $O/err_$w.txt
done
python - <<'PY'
import json
for w in ("cfg3_dragon250k","cfg2_bunny70k","scroll250k"):
    for suf in ("","_longest"):
        d=json.loads(open(f"gpurun_out/r05_l/bench_{w}{suf}.json").read().strip().splitlines()[-1]); c=d["config"]
        print(w+suf, round(d["ms_per_step"],4), c["dissection"], c["factor_seconds"], c["factor_seconds_steady"], c["tolerance"]["measured"])
PY
for w in cfg3_dragon250k cfg2_bunny70k; do timeout 600 python tools/bench_remesh.py $w 100 6 2>&1 | grep -v amdgpu.ids; LS_ND_ORDER=0 timeout 600 python tools/bench_remesh.py $w 100 6 2>&1 | grep -v amdgpu.ids | sed 's/^/[LS_ND_ORDER=0] /'; done | tee $O/remesh.txt
