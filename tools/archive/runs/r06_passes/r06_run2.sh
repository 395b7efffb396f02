set -x
mkdir -p gpurun_out/r2
python tools/irregular_1m.py 200 --table > gpurun_out/r2/irregular.txt 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "irregular or sphere_vs_oracle or direct_options or ordering_argument or sixteen_wave or one_million or cfg4 or cfg5" > gpurun_out/r2/pytest_new.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2/pytest_new.log
python bench.py --no-extra-baselines --workload cfg4b_sphere1m > gpurun_out/r2/bench_sphere.json 2> gpurun_out/r2/bench_sphere.err
tail -3 gpurun_out/r2/pytest_new.log
cat gpurun_out/r2/irregular.txt
