set -x
mkdir -p gpurun_out/r4
python tools/irregular_1m.py 300 --quick --table > gpurun_out/r4/quick.txt 2>&1
LARGESTEPS_HIP_LIB=$PWD/tools/build/v_stamps/liblargesteps_hip.so python tools/tier_stamps.py cfg4_plane1m > gpurun_out/r4/stamps_plane16.txt 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_nested_gpu.py -m gpu -x -q -k "irregular or sixteen_wave or one_million or cfg1 or cfg4 or columns or nested or direct" > gpurun_out/r4/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4/pytest.log
tail -3 gpurun_out/r4/pytest.log
cat gpurun_out/r4/quick.txt gpurun_out/r4/stamps_plane16.txt
