for o in -1 1; do LS_ND_ORDER=$o python bench.py --steps 50 --warmup 5 --workload cfg3_dragon250k --no-cpu-baseline 2>/dev/null > gpurun_out/cfg3_$o.json; done
python - <<'PY'
import json
for o in ("-1", "1"):
    d = json.loads(open(f"gpurun_out/cfg3_{o}.json").read().strip().splitlines()[-1]); c = d["config"]
    print("LS_ND_ORDER", o, round(d["ms_per_step"], 4), c["dissection"], c["factor_seconds_steady"], c["factor_seconds_cycles"])
PY
