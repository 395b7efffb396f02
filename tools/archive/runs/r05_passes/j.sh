cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_j
timeout 200 python bench.py --steps 50 --warmup 3 --reorder --no-cpu-baseline > gpurun_out/r05_j/bench_reordered.json 2> gpurun_out/r05_j/err.txt; echo "rc=$?"
timeout 200 python bench.py --steps 50 --warmup 3 --no-cpu-baseline > gpurun_out/r05_j/bench_plain.json 2>> gpurun_out/r05_j/err.txt; echo "rc=$?"
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k reorder 2>&1 | tail -3
python - <<'PY'
import json
for f in ("bench_reordered","bench_plain"):
    d=json.loads(open(f"gpurun_out/r05_j/{f}.json").read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["config"]["kernel_us"], d["config"]["max_abs_err_vs_v"])
PY
