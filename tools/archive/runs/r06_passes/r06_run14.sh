set -x
mkdir -p gpurun_out/r14
O=gpurun_out/r14
python -m pytest tests/test_gpu_parity.py tests/test_nested_gpu.py -m gpu -x -q -k "irregular or sixteen_wave or one_million or cfg or columns or nested or direct or sphere or cache_policy" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python tools/irregular_1m.py 300 --quick --table > $O/quick_ksplit.txt 2>&1
LS_ND_KSPLIT=0 python tools/irregular_1m.py 300 --quick --table > $O/quick_noksplit.txt 2>&1
tail -3 $O/pytest.log; grep -v amdgpu $O/quick_ksplit.txt; grep -v amdgpu $O/quick_noksplit.txt
