cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_i
timeout 600 python bench.py --steps 50 --warmup 3 --reorder --no-extra-baselines > gpurun_out/r05_i/bench_reordered.json 2> gpurun_out/r05_i/bench_reordered.err; echo "rc=$?"
tail -20 gpurun_out/r05_i/bench_reordered.err; cut -c1-300 gpurun_out/r05_i/bench_reordered.json
