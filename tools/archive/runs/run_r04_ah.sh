#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04_ah; rm -rf $O; mkdir -p $O
timeout 600 python bench.py --steps 50 --warmup 3 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_ah/bench.json').read().splitlines() if l.startswith('{')][-1])
c=d['config']; print(d['ms_per_step'], c['factor_seconds'], c['factor_seconds_second_construction'], c['factor_seconds_steady'])
PY
tail -2 $O/bench.err
