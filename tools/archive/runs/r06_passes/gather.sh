mkdir -p gpurun_out/gather
for g in 1 0 1 0; do
  echo "== LS_ND_GATHER=$g"; LS_ND_GATHER=$g python tools/irregular_1m.py 300 --quick 2>&1 | grep -v amdgpu | head -3
done > gpurun_out/gather/out.txt
echo "== default rule" >> gpurun_out/gather/out.txt; python tools/irregular_1m.py 300 --quick 2>&1 | grep -v amdgpu | head -3 >> gpurun_out/gather/out.txt
python -m pytest tests/test_gpu_parity.py tests/test_nested_gpu.py -m gpu -x -q -k "irregular or sixteen_wave or one_million or cfg1 or cfg4 or columns or nested or direct or sphere" > gpurun_out/gather/pytest.log 2>&1; tail -2 gpurun_out/gather/pytest.log
LS_ND_GATHER=1 python -m pytest tests/test_gpu_parity.py tests/test_nested_gpu.py -m gpu -x -q -k "sixteen_wave or one_million or cfg1 or cfg4 or columns or nested or direct" > gpurun_out/gather/pytest_forced.log 2>&1; tail -2 gpurun_out/gather/pytest_forced.log
cat gpurun_out/gather/out.txt
